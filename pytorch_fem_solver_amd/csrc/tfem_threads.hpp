// Host-side threading of the once-per-mesh builders (symbolic phase, plans): plain std::thread,
// static partition.  TFEM_HOST_THREADS overrides the count (1 = sequential; default: the
// hardware concurrency, at most 16 -- the CPU share of one GPU on the target boxes).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace tfem {

inline int host_threads() {
  if (const char *v = std::getenv("TFEM_HOST_THREADS")) {
    const int n = std::atoi(v);
    if (n >= 1) return std::min(n, 256);
  }
  const unsigned hw = std::thread::hardware_concurrency();
  return int(std::max(1u, std::min(hw ? hw : 1u, 16u)));
}

// fn(begin, end, thread_index) over [0, n) cut into one contiguous piece per thread.
template <typename F>
void parallel_for(int64_t n, F fn, int64_t min_per_thread = 1024) {
  const int64_t want = std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / std::max<int64_t>(min_per_thread, 1)));
  if (want <= 1) {
    fn(int64_t(0), n, 0);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve(size_t(want));
  for (int64_t t = 0; t < want; ++t) {
    const int64_t b = n * t / want, e = n * (t + 1) / want;
    pool.emplace_back([=, &fn]() { fn(b, e, int(t)); });
  }
  for (std::thread &th : pool) th.join();
}

}  // namespace tfem
