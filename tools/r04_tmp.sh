#!/bin/bash
bash tools/r04_quick.sh gpurun_out/r04q6 || exit 1
