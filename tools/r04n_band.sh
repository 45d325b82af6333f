# round 4: fp32 band attention, workgroup-staged (band_attn_f32.hip): lab sweep
python sl-hwgat_amd/build.py > /dev/null 2>&1; echo "build rc $?"
python sl-hwgat_amd/build.py --lab > /dev/null 2>&1; echo "lab build rc $?"
mkdir -p gpurun_out/r04n
timeout -k 10 300 python tools/band_f32_lab.py > gpurun_out/r04n/band_f32_lab.txt 2>&1; echo "rc $?"; cat gpurun_out/r04n/band_f32_lab.txt
