# what the GPU box's host gives a process: logical CPUs, SMT numbering, cgroup CPU quota / throttling, and the speed of ONE 8-thread CPU-baseline worker alone
cd $GRAFT_REPO_ROOT
O=gpurun_out/hostcpu.txt
{
echo "# nproc: $(nproc)   affinity: $(python3 -c 'import os; print(len(os.sched_getaffinity(0)))')"
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|MHz" 
echo "# siblings of cpu0: $(cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list)   of cpu1: $(cat /sys/devices/system/cpu/cpu1/topology/thread_siblings_list)"
echo "# cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)   (v1 quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null))"
echo "# cpu.stat before:"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
echo "# memory: $(grep MemTotal /proc/meminfo)"
python3 tools/debug/cpu_workers_probe.py 1x8 1x16 4x8 16x8 16x16 2>&1 | tail -6
echo "# cpu.stat after:"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
} > $O 2>&1
cat $O
