// pf_host.h -- internal: how many host threads a parallel section of the library may start.
// Not part of the C ABI (include/panfeed_hip.h).
#pragma once

// std::thread::hardware_concurrency() says what the MACHINE has (256 on a GPU box whose container is given 16 CPUs by a
// CFS quota).  = min(hardware_concurrency, the affinity mask, TWICE the cgroup's cpu quota), at least 1; read once.
// Twice: measured on that box (round 5) the latency-bound sections are faster with 32 threads than with 16 -- more cache
// misses in flight -- and the rest indifferent; the quota still bounds a library dropped into a 2-CPU container.
// `cap`: the section's own upper bound.
unsigned pf_host_threads(unsigned cap);
