mkdir -p gpurun_out
for lib in "$@"; do
  echo "== $lib"
  DW_LIB=isaacgymdyros_amd/_ab/$lib python tools/pipe_time.py --pipes 3 --envs 4096,16384 --rounds 3 2>&1 | grep "N="
done
