# A/B of two library builds on ONE box (boxes differ by several per cent in clocks):
#   1. build variant A, `cp tacotron2-vae_amd/libt2vae_hip.so gpurun_in/libA.so`; build variant B (stays in place)
#   2. gpurun -- 'bash tools/dbg/ab_lib.sh gpurun_in/libA.so "<command>"'   -> runs <command> with A, B, A, B
LIBA=$1; shift
for i in 1 2; do
  echo "== A ($LIBA)"; T2V_LIB=$(pwd)/$LIBA bash -c "$*"
  echo "== B (in-tree)"; bash -c "$*"
done
