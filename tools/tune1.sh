for v in nt1_mw1 nt0_mw1 nt1_mw2 nt0_mw2 nt1_mw4 nt0_mw4; do
  DFX_LIB=$PWD/build/libdfx_$v.so timeout -k 10 120 python -u tools/tune_kernels.py --zc 16,8,4 --steps 32,64 >> gpurun_out/tune1.txt 2>&1 || exit 1
done
