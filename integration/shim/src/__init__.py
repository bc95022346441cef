"""Shim package: put `integration/shim` ahead of the reference checkout on sys.path and the reference's own scripts (demo.py, demo_video.py,
src/smirk_trainer.py) import the MI355X drop-in classes through their usual `src.*` module paths."""
