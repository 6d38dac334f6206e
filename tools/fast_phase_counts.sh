for n in 1 2 3 4 5 6 0; do
  echo -n "stop $n: "
  ORBGPU_FAST_STOP=$n ORBGPU_LIBRARY=$PWD/orb_slam3_amd/liborbgpu_prof.so bash tools/quick_pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "k_fast_cells<false,0>" | tr '\n' ' '; echo
done
