# round 4: wave-cycle / wait / LDS-conflict / MFMA-busy counters of the four-wave fp32 block attention kernels
python sl-hwgat_amd/build.py > /dev/null 2>&1; echo "build rc $?"
PMC_OUT=r04m PMC_ARGS=f32 PMC_FILTER=blk_ bash tools/blk_pmc.sh
