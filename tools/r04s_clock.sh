# round 4: shader clock under the fp32 NT linear kernel (lab library)
python sl-hwgat_amd/build.py > /dev/null 2>&1; echo "build rc $?"
python sl-hwgat_amd/build.py --lab > /dev/null 2>&1; echo "lab build rc $?"
mkdir -p gpurun_out/r04s
timeout -k 10 300 python tools/nt256_clock.py > gpurun_out/r04s/nt256_clock.txt 2>&1; echo "rc $?"; cat gpurun_out/r04s/nt256_clock.txt
