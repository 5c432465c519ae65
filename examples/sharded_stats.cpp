// sharded_stats.cpp — north_star config 5 through the C++ host layer (host/arrow_gpu.hpp) alone: no Python, no torch.
// One host thread per visible MI355X; thread r owns GpuDevice r, a pipeline and rank r of an RCCL communicator; the
// column has world × rows_per_gpu rows, shard r = rows [r·rows, (r+1)·rows) generated on its GPU by the counter-based
// hash (agpu_synth_f32: any shard of a column can be generated anywhere).  Every rank computes whole-column
// Sum (reference tree order) / min / max with sum_sharded / min_sharded / max_sharded and must see the SAME bits.
// Not in the reference (single device, crates/array/src/gpu_utils/gpu_device.rs:29-33).
//
//   hipcc -std=c++17 -O2 -x c++ examples/sharded_stats.cpp -o sharded_stats -Larrow_gpu_amd/lib -larrow_gpu_hip
//   ./sharded_stats [rows_per_gpu] [world]        → one JSON line
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../host/arrow_gpu.hpp"

using namespace arrow_gpu;

struct RankResult {
  float sum = 0, mn = 0, mx = 0;
  double ms = 0;
  std::string error;
};

int main(int argc, char** argv) {
  const uint64_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
  int32_t ndev = 0;
  agpu_device_count(&ndev);
  if (ndev <= 0) {
    printf("no device: this program has no CPU fallback\n");
    return 2;
  }
  const int world = argc > 2 ? atoi(argv[2]) : ndev;
  if (world < 1 || world > ndev) {
    printf("world must be 1..%d (one rank per GPU)\n", ndev);
    return 2;
  }
  const Communicator::Id id = Communicator::unique_id();
  std::vector<RankResult> res((size_t)world);
  auto rank_main = [&](int r) {
    try {
      auto dev = GpuDevice::create(r);
      ArrowComputePipeline p(dev, "rank");
      Communicator comm(dev, id, r, world);  // collective: all threads arrive here
      const Shard sh = shard_rows(rows * (uint64_t)world, world, r);
      auto buf = dev->create_empty_buffer(sh.rows * 4);
      check(agpu_synth_f32(p.h(), (float*)buf->ptr, sh.rows, 20250418, sh.row0, -1.0f, 1.0f), "agpu_synth_f32");
      Float32ArrayGPU shard(buf, dev, sh.rows, std::nullopt);
      (void)sum_sharded_op(shard, comm, p);  // warm-up: scratch + RCCL's first call
      p.sync();
      comm.barrier(p);
      const auto t0 = std::chrono::steady_clock::now();
      auto s = sum_sharded_op(shard, comm, p);
      auto lo = min_sharded_op(shard, comm, p);
      auto hi = max_sharded_op(shard, comm, p);
      p.sync();
      comm.barrier(p);
      res[(size_t)r].ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      res[(size_t)r].sum = s.raw_values()[0];
      res[(size_t)r].mn = lo.raw_values()[0];
      res[(size_t)r].mx = hi.raw_values()[0];
    } catch (const std::exception& e) {
      res[(size_t)r].error = e.what();
    }
  };
  std::vector<std::thread> ts;
  for (int r = 1; r < world; r++) ts.emplace_back(rank_main, r);
  rank_main(0);
  for (auto& t : ts) t.join();
  bool ok = true;
  double ms = 0;
  for (int r = 0; r < world; r++) {
    if (!res[(size_t)r].error.empty()) {
      printf("rank %d failed: %s\n", r, res[(size_t)r].error.c_str());
      ok = false;
    }
    ok = ok && memcmp(&res[(size_t)r].sum, &res[0].sum, 4) == 0 && res[(size_t)r].mn == res[0].mn && res[(size_t)r].mx == res[0].mx;
    if (res[(size_t)r].ms > ms) ms = res[(size_t)r].ms;
  }
  printf("{\"what\": \"C++ host: chunk-sharded f32 sum/min/max, RCCL final reduce\", \"world\": %d, \"rows_per_gpu\": %llu, "
         "\"sum\": %.9g, \"min\": %.9g, \"max\": %.9g, \"ms_3_statistics\": %.4f, \"aggregate_GBps\": %.1f, "
         "\"identical_on_all_ranks\": %s}\n",
         world, (unsigned long long)rows, res[0].sum, res[0].mn, res[0].mx, ms, 3.0 * 4.0 * (double)rows * world / ms / 1e6,
         ok ? "true" : "false");
  return ok ? 0 : 1;
}
