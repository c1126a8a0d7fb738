#!/bin/bash
export MNET_GIT_COMMIT=83b43ed
bash tools/round_profiles.sh r4w fp16x2 2>&1 | tail -5
