// The threads this process may keep busy: the hardware's, cut to the CPUs the process may run on (its affinity mask) and to the
// cgroup's CPU quota.  A container that sees 256 hardware threads under a quota of 16 CPUs (the GPU boxes of this project:
// /sys/fs/cgroup/cpu.max = "1600000 100000", profiles/r06_notes.md) is stopped for the rest of every 100 ms period once its threads
// have used 1.6 s of CPU time in it: 64 walkers beside 48 rendering threads stood still for 60-90 ms at a time.  HGX_HOST_THREADS
// overrides.
#pragma once
#include <sched.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace hgx {

inline unsigned hostThreads() {
    static const unsigned threads = []() {
        if (const char *e = getenv("HGX_HOST_THREADS"))
            return (unsigned)std::max(1, atoi(e));
        unsigned n = std::thread::hardware_concurrency();
        if (n == 0)
            n = 1;
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
            n = std::min(n, (unsigned)CPU_COUNT(&set));
        // cgroup v2: "<quota> <period>" or "max <period>"; v1: two files
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0)
                quota = atoll(q);
            fclose(f);
        } else {
            if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(fq, "%lld", &quota) != 1)
                    quota = -1;
                fclose(fq);
            }
            if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(fp, "%lld", &period) != 1)
                    period = 0;
                fclose(fp);
            }
        }
        if (quota > 0 && period > 0)
            n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        return n;
    }();
    return threads;
}

// For a phase of a few milliseconds: a quota is an amount of CPU time per period (1.6 CPU-seconds per 100 ms on the GPU boxes), not a
// number of threads — 64 threads for 6 ms spend a quarter of a period's allowance and finish in a third of the time 16 take
// (hgx_liftover_convert of a million lines on the box: 6.3 ms with 64 threads, 11.1 with 16: profiles/r06_notes.md).  Up to four
// times the quota's CPUs, the hardware and the affinity mask permitting; work that goes on for tenths of a second keeps to hostThreads().
inline unsigned hostBurstThreads() {
    static const unsigned threads = []() {
        if (getenv("HGX_HOST_THREADS"))
            return hostThreads();
        unsigned n = std::thread::hardware_concurrency();
        if (n == 0)
            n = 1;
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
            n = std::min(n, (unsigned)CPU_COUNT(&set));
        return std::max(hostThreads(), std::min(n, 4u * hostThreads()));
    }();
    return threads;
}

} // namespace hgx
