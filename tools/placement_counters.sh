# Counter passes for the two placement regimes (DESIGN.md "Placement of large vectors"): the same sweep, x in the
# first 16 GiB of one 240 GiB arena, p at +32 GiB (slow regime) and at +96 GiB (fast regime), 3 launches each.
# One rocprofv3 --pmc run per pass (<= 4 TCC, 4 TCP, 2 GRBM counters), the program itself after `--`.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/placement_counters
rm -rf $OUT; mkdir -p $OUT
OFFS=${1:-32,96}
pass() {  # name, counters...
  n=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" --kernel-trace --output-format csv json -d $OUT/$n -- python3 tools/placement_counters.py --offsets-gib $OFFS --reps 3 > $OUT/$n.txt 2>&1
}
pass p1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY
pass p2 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_EA_BUSY GRBM_TC_BUSY
pass p3 TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_sum
pass p4 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_GMI_32B_sum
pass p5 TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ
python3 tools/placement_counters_report.py $OUT $OFFS | tee $OUT/report.txt
