set -e
O=gpurun_out/tune7.txt; rm -f $O
T="timeout -k 10 280 python -u tools/tune_kernels.py"
P="-1,0,4096,69632,1052672,2101248,33558528"
$T --shape 256,16,4,512,512 --zc 4 --steps 64 --blocks 256 --rounds 4 --pad=$P >> $O 2>&1
$T --shape 512,32,4,256,256 --zc 4 --steps 64 --blocks 256 --rounds 4 --pad=$P >> $O 2>&1
$T --zc 8 --steps 64 --blocks 256 --pad=$P >> $O 2>&1
