"""TEST INFRASTRUCTURE ONLY — CPU oracle for the SMIRK per-frame hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker.  The product path (``smirk_amd``) never imports this
package and fails loudly when the HIP library is missing.

Contents
  assets.py        synthetic-asset sandbox (FLAME pkl, obj, masks, embeddings) built from
                   tests/golden/assets_bundle.npz so it works where /root/reference is absent
  sandbox.py       imports the *real* reference modules (only where /root/reference exists)
                   with the third-party shims of SURVEY.md App. D; used to pin the restatements
  flame_ref.py     numpy restatement of src/FLAME/{FLAME,lbs}.py
  render_ref.py    numpy restatement of src/renderer/{renderer,util}.py (+ raster_ref.c)
  raster_ref.c     plain-C restatement of pytorch3d RasterizeMeshesNaiveCpu (parity unpinned)
  generator_ref.py torch-CPU fp32 functional restatement of src/smirk_generator.py
  mobilenet_ref.py torch-CPU fp32 restatement of timm tf_mobilenetv3_*_minimal_100 features
                   (parity unpinned: timm is not on disk) + src/smirk_encoder.py heads
"""
