#!/bin/bash
# phase clock of the shadow kernels (prof build) on dodge and cfg4
mkdir -p gpurun_out/phase
timeout -k 10 200 python tools/prof.py dodgeColorTest.obj 1920 1080 8 4 > gpurun_out/phase/dodge.log 2>&1 || { tail -5 gpurun_out/phase/dodge.log; exit 1; }
grep -E "by phase|shaft walk" gpurun_out/phase/dodge.log
timeout -k 10 300 python tools/prof.py wavy 3840 2160 16 8 > gpurun_out/phase/wavy.log 2>&1 || { tail -5 gpurun_out/phase/wavy.log; exit 1; }
grep -E "by phase|shaft walk" gpurun_out/phase/wavy.log
