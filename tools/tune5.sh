set -e
O=gpurun_out/tune5.txt; rm -f $O
T="timeout -k 10 200 python -u tools/tune_kernels.py"
for il in 0 1; do
export DFX_LIB=$PWD/build/libdfx_il$il.so
echo "== interleave $il" >> $O
$T --zc 8,4 --steps 64 --blocks 256,512,1024 >> $O 2>&1
$T --shape 512,32,4,256,256 --zc 4 --steps 64 --blocks 256,512,1024 --rounds 4 >> $O 2>&1
$T --shape 256,16,4,512,512 --zc 4 --steps 64 --blocks 256,512 --rounds 4 >> $O 2>&1
$T --shape 64,16,20,128,128 --zc 4 --steps 64 --blocks 256,512 --rounds 8 >> $O 2>&1
done
