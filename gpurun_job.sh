mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in mix d2000; do
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_psd_$w -o psd -- python /root/repo/tools/psd_profile_workload.py $w 2>&1 | grep -E "^(mix|d200|d2000)"
done
ls /root/repo/gpurun_out/pmc_psd_mix/
