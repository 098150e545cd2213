#!/bin/bash
# usage: tools/isa_build.sh <name> [extra hipcc flags]: listing of the octet unit with line tables -> /tmp/isa/<name>.s, then the register / scratch
# usage of its kernels and the estimated dynamic instruction counts of the flat two-waves step kernel (tools/isa_dyn.py)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p /tmp/isa
FLAGS=$(python -c "from isaacgymdyros_amd import build; print(' '.join(build.FLAGS + dict(build.SOURCES)['dw_oct_kernels.hip']))")
(cd isaacgymdyros_amd/csrc && hipcc $FLAGS -I../../include "$@" -gline-tables-only -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -o /tmp/isa/$NAME.s dw_oct_kernels.hip 2>&1 \
  | grep -E "Function Name|ScratchSize|VGPRs:|AGPRs" | sed 's/\[-Rpass.*//' | paste - - - - | sed "s/dw_oct_kernels.hip:[0-9]*:1: remark: //g" | sed 's/EvPKN3dwq.*\] */ /')
python tools/isa_dyn.py /tmp/isa/$NAME.s dw_k_step_octILb0ELi2ELi1 | head -31
