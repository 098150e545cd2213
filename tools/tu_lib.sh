#!/bin/bash
# usage: tools/tu_lib.sh <name> <unit.hip> [hipcc flags that REPLACE OR EXTEND the unit's flags ...]
# Builds a complete library in which ONE translation unit is compiled with extra flags into isaacgymdyros_amd/_ab/<name>.so; the
# other units are the objects of the in-tree build (isaacgymdyros_amd/_obj).  The unit's base flags come from
# isaacgymdyros_amd/build.py (FLAGS + the unit's own), so a variant differs from the product by exactly the flags given here;
# DW_TU_DROP="-fno-slp-vectorize ..." removes base flags.  Tools only -- the product always loads the in-tree library.
set -e
cd "$(dirname "$0")/.."
NAME=$1; UNIT=$2; shift 2
mkdir -p isaacgymdyros_amd/_ab
[ -f isaacgymdyros_amd/_obj/dw_hip.o ] || python -c "from isaacgymdyros_amd import build; build.build(force=True)"
BASE=$(python - "$UNIT" <<'PY'
import sys
from isaacgymdyros_amd import build
u = sys.argv[1]
print(" ".join(build.FLAGS + dict(build.SOURCES)[u]))
PY
)
for d in ${DW_TU_DROP:-}; do BASE=${BASE//$d/}; done
STEM=${UNIT%.hip}
hipcc $BASE -Iinclude "$@" -Rpass-analysis=kernel-resource-usage -c -o isaacgymdyros_amd/_ab/${NAME}_${STEM}.o isaacgymdyros_amd/csrc/$UNIT 2>&1 \
  | grep -E "error|Function Name|ScratchSize|VGPRs:|AGPRs" | sed 's/\[-Rpass.*//' | paste - - - - | sed "s/$UNIT:[0-9]*:1: remark: //g" || true
OBJS=""
for s in $(python -c "from isaacgymdyros_amd import build; print(' '.join(s for s, _ in build.SOURCES))"); do
  if [ "$s" = "$UNIT" ]; then OBJS="$OBJS isaacgymdyros_amd/_ab/${NAME}_${STEM}.o"; else OBJS="$OBJS isaacgymdyros_amd/_obj/${s%.hip}.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o isaacgymdyros_amd/_ab/${NAME}.so $OBJS
echo isaacgymdyros_amd/_ab/${NAME}.so
