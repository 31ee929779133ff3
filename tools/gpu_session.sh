#!/bin/bash
bash tools/profile_round.sh gpurun_out/r6q
