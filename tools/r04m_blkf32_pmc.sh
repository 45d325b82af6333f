python sl-hwgat_amd/build.py > /dev/null 2>&1; echo "build rc $?"
python sl-hwgat_amd/build.py --lab > /dev/null 2>&1; echo "lab build rc $?"
mkdir -p gpurun_out/r04m
timeout -k 10 120 python tools/blk_stamps.py > gpurun_out/r04m/blk_stamps.txt 2>&1; echo "rc $?"; cat gpurun_out/r04m/blk_stamps.txt
