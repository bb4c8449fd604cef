set -e
O=gpurun_out/tune6.txt; rm -f $O
T="timeout -k 10 200 python -u tools/tune_kernels.py"
for v in nov il1; do
export DFX_LIB=$PWD/build/libdfx_$v.so
$T --shape 512,32,4,256,256 --zc 4 --steps 64 --blocks 256,512 --rounds 4 >> $O 2>&1
$T --shape 256,16,4,512,512 --zc 4 --steps 64 --blocks 256,512 --rounds 4 >> $O 2>&1
$T --zc 8 --steps 64 --blocks 256 >> $O 2>&1
done
