# CPU-baseline worker shapes under the box's cgroup CPU quota (cpu.max 1600000/100000 = 16 CPUs of time for 256 visible logical CPUs)
cd $GRAFT_REPO_ROOT
{
echo "# cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
python3 tools/debug/cpu_workers_probe.py ${SHAPES:-2x8 3x8 4x8 5x8 6x8 8x8 4x12 4x16 8x4 6x6 16x2 16x1 4x8}
} > gpurun_out/hostcpu2.txt 2>&1
cat gpurun_out/hostcpu2.txt
