set -e
O=gpurun_out/tune8.txt; rm -f $O
export TUNE_VERBOSE=1
T="timeout -k 10 280 python -u tools/tune_kernels.py"
P="0,4096,0,4096,1052672,2101248,1052672,2101248,8192,16384,32768,65536,131072,262144,524288"
$T --shape 256,16,4,512,512 --zc 4 --steps 64 --blocks 256 --rounds 6 --pad=$P >> $O 2>&1
