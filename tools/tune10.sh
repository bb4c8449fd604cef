set -e
O=gpurun_out/tune10.txt; rm -f $O
T="timeout -k 10 280 python -u tools/tune_kernels.py"
for n in 8 16 32; do
export DFX_LIB=$PWD/build/libdfx_bi$n.so
$T --zc 8,4 --steps 64 --blocks 256 >> $O 2>&1
$T --shape 512,32,4,256,256 --zc 4 --steps 64 --blocks 256 --rounds 4 >> $O 2>&1
$T --shape 64,16,20,128,128 --zc 4 --steps 64 --blocks 256,512 --rounds 8 >> $O 2>&1
done
