"""Encoder output tolerances, derived from measurement instead of round-2 guesses (tools/encoder_error_table.py on the MI355X, profiles/r05a_encoder_error_table.txt:
192 frames of the B = 128 / 256 / 1024 batches of tests/test_scale_gpu.py, synthetic BN-calibrated weights).  Max |error| per head:

    head                 |value|   HIP vs float64   torch-CPU fp32 vs float64   HIP vs torch-CPU fp32
    pose_params            0.66       5.7e-6              5.1e-6                     2.6e-6
    cam                   10.8        3.5e-5              2.8e-5                     1.1e-5
    shape_params           3.8        1.6e-4              1.3e-4                     4.0e-5
    expression_params      6.5        2.9e-4              2.8e-4                     1.2e-4
    eyelid_params          1.0        2.9e-5              3.3e-5                     1.4e-5
    jaw_params             0.95       2.1e-5              1.9e-5                     6.7e-6

The HIP path (split-fp16 x3 MFMA + fp32 streaming kernels) is as far from float64 as torch's own CPU fp32 evaluation is.  The loosest head, expression_params, is head
amplification of fp32 rounding in the pooled features, not a layer's error: the 960 pooled features carry 2.1e-5 rms error at |f| <= 50 (4e-7 relative), the calibrated
Linear(960 -> 55) has row L2 norms up to 28 -> 5.9e-4 expected for independent errors, 1.2e-4 .. 2.9e-4 observed.

VS_FP32: 4 x max |HIP - torch-CPU fp32| rounded to one digit (what a comparison against the fp32 oracle / reference golden can see);
VS_FP64: 4 x max |HIP - float64| (comparisons against the float64 oracle, and against fp32 references on frames other than the measured ones).
Round 5 shipped 2 x the sample maximum; the maxima come from 192 frames of ONE box, one ROCm build, one host BLAS (the fp32 oracle's own rounding moves with the CPU's
oneDNN / BLAS build), so a 2 x bound makes the suite flaky without any library bug (advisor, round 5).  4 x stays below the error model's one-sigma figure for the loosest head
(5.9e-4 expected for independent errors on expression_params) on the fp32 side and 3.3x - 16x tighter than round 4's 2e-4 ... 1e-3 guesses everywhere else."""
VS_FP32 = dict(pose_params=1.1e-5, cam=5e-5, shape_params=1.6e-4, expression_params=5e-4, eyelid_params=6e-5, jaw_params=3e-5)
VS_FP64 = dict(pose_params=2.3e-5, cam=1.4e-4, shape_params=6.4e-4, expression_params=1.2e-3, eyelid_params=1.2e-4, jaw_params=8.4e-5)
