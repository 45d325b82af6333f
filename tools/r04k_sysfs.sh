# what the GPU box's sysfs shows an ordinary user (for bench.gpu_local_cpus)
O=gpurun_out/r04k
mkdir -p $O
{
ls -la /sys/class/kfd/kfd/topology/nodes/ 2>&1 | head -30
for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "== $n"; cat $n/properties 2>&1 | grep -E "simd_count|location_id|domain|cpu_cores_count|drm_render_minor|unique_id" ; done
echo "HIP_VISIBLE_DEVICES=$HIP_VISIBLE_DEVICES ROCR_VISIBLE_DEVICES=$ROCR_VISIBLE_DEVICES CUDA_VISIBLE_DEVICES=$CUDA_VISIBLE_DEVICES"
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
print("gpu_local_cpus(0):", bench.gpu_local_cpus(0))
print("visible_gpu_count:", bench.visible_gpu_count())
print("affinity:", len(os.sched_getaffinity(0)))
import glob
for p in glob.glob("/sys/bus/pci/devices/*/local_cpulist")[:3]: print(p, open(p).read().strip())
PY
} > $O/sysfs.txt 2>&1
cat $O/sysfs.txt | head -80
