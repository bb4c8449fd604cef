set -e
O=gpurun_out/tune12.txt; rm -f $O
T="timeout -k 10 280 python -u tools/tune_kernels.py"
for n in 1 2 4; do
export DFX_LIB=$PWD/build/libdfx_zb$n.so
$T --zc 8,4 --steps 64 >> $O 2>&1
$T --shape 64,16,20,128,128 --zc 4 --steps 64 --rounds 8 >> $O 2>&1
done
