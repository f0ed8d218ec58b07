#!/bin/bash
# A/B of a build-time switch of csrc/ on ONE box: the default library, then one built with the given define, then the default again
# (drift between boxes and over minutes is larger than most effects here).  usage: ab_build_define.sh -DAA_EARLY_RIDE_OUT=0 [reps]
DEF=$1; REPS=${2:-3}
leg() {
    echo "== build defines: '${NAF_BUILD_DEFINES:-}'"
    for i in $(seq $REPS); do
        python benchmarks/host_api_steps.py 64 2>/dev/null | grep -o "batch [0-9]* .*path: [0-9]* timesteps/s"
        python benchmarks/host_api_steps.py 256 2>/dev/null | grep -o "batch [0-9]* .*path: [0-9]* timesteps/s"
    done
}
unset NAF_BUILD_DEFINES; leg
export NAF_BUILD_DEFINES="$DEF"; leg
unset NAF_BUILD_DEFINES; python -c "from robotic_manipulator_rloa_amd import _lib; _lib.build_library(force=True)"; leg
