set -e
O=gpurun_out/tune4.txt; rm -f $O
T="timeout -k 10 200 python -u tools/tune_kernels.py"
for w in 4 8; do
export DFX_LIB=$PWD/build/libdfx_w$w.so
echo "== waves $w C2" >> $O
$T --zc 8,4 --steps 64 --blocks 128,256,512 >> $O 2>&1
echo "== waves $w c3g8-like" >> $O
$T --shape 512,32,4,256,256 --zc 4 --steps 64 --blocks 128,256,512 --rounds 4 >> $O 2>&1
done
unset DFX_LIB
export DFX_LIB=$PWD/build/libdfx_w4.so
echo "== A: c3 z16 with zc 4 / 8; nZ=4 at 128^2" >> $O
$T --shape 512,32,16,256,256 --zc 4,8 --steps 64 --blocks 256,1024 --rounds 3 >> $O 2>&1
$T --shape 1024,16,4,128,128 --zc 4 --steps 64 --blocks 256,1024 --rounds 4 >> $O 2>&1
$T --shape 256,16,4,256,256 --zc 4 --steps 64 --blocks 256,1024 --rounds 4 >> $O 2>&1
