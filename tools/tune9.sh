set -e
O=gpurun_out/tune9.txt; rm -f $O
export TUNE_ARENA=1
T="timeout -k 10 280 python -u tools/tune_kernels.py"
P="0,4096,8192,16384,32768,65536,131072,262144,524288,1048576,2097152,4194304,8388608,16777216,33554432,0,4096,12288,20480,1052672,3145728"
$T --shape 256,16,4,512,512 --zc 4 --steps 64 --blocks 256 --rounds 4 --pad=$P >> $O 2>&1
$T --shape 256,16,4,512,512 --zc 4 --steps 64 --blocks 256 --rounds 4 --pad=$P >> $O 2>&1
