#!/bin/bash
# raster_tile: how much of it is the per-tile binning scan (ablate 2 skips it), how much the per-pixel face loop (ablate 1 skips it)?
cd /root/repo
for a in 0 1 2 3; do SMIRK_RASTER_ABLATE=$a python tools/raster_time.py 2>&1 | tail -1; done
