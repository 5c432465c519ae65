// sharded_stats.cpp — north_star config 5 AND the headline step through the C++ host layer (host/arrow_gpu.hpp) alone: no
// Python, no torch, ONE process.  One host thread per visible MI355X; thread r owns GpuDevice r, a pipeline and rank r of
// an RCCL communicator (the id is passed between the threads directly).  The column is chunk-sharded:
//   weak   : world × rows rows, shard r = rows [r·rows, (r+1)·rows)
//   strong : rows rows in total, cut by shard_rows into `world` contiguous shards (125 M per GPU at 1e9 rows / 8 GPUs)
// generated on each GPU by the counter-based hash (agpu_synth_*: any shard of a column can be generated anywhere).
// Timed, bracketed by a communicator barrier + stream sync on both sides, MAX over ranks:
//   (1) `steps` steps of  f32 add  +  i32 eq → bitmap with fused validity AND   (bench.py's step: 12 + 8.5 B/row),
//       per-launch HIP-event means per rank (a straggler is visible as min ≠ max);
//   (2) whole-column Sum (reference tree order) / min / max with sum_sharded / min_sharded / max_sharded — every rank must
//       see the SAME bits.
// The in-process twin of `torchrun … bench.py`: if the two disagree at N > 1, the difference is the process model, not the
// kernels.  Not in the reference (single device, crates/array/src/gpu_utils/gpu_device.rs:29-33).
//
//   hipcc -std=c++17 -O2 -x c++ examples/sharded_stats.cpp -o sharded_stats -Larrow_gpu_amd/lib -larrow_gpu_hip -lpthread
//   ./sharded_stats [rows] [world] [weak|strong] [steps]        → one JSON line
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../host/arrow_gpu.hpp"

using namespace arrow_gpu;

struct RankResult {
  float sum = 0, mn = 0, mx = 0;
  double stats_ms = 0, step_s = 0, add_ms = 0, eq_ms = 0, one_pass_ms = 0;
  bool one_pass_same = false;
  uint64_t rows = 0;
  int rccl_ranks = 0, distinct_devices = 0;
  std::string error;
};

int main(int argc, char** argv) {
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);  // dmabuf-only hosts: must be in place before the first HIP call (a default, not an override)
  // every rank of this program is a thread of THIS process: RCCL's rendezvous socket goes over the loopback interface, the one a container
  // cannot firewall or rename (a default, not an override; data travels over xGMI / shared memory either way)
  setenv("NCCL_SOCKET_IFNAME", "lo", 0);
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
  const bool strong = argc > 3 && std::string(argv[3]) == "strong";
  const int steps = argc > 4 ? std::max(1, atoi(argv[4])) : 20;
  const uint64_t total = strong ? rows : rows * (uint64_t)world;
  const Communicator::Id id = Communicator::unique_id();
  std::vector<RankResult> res((size_t)world);
  auto rank_main = [&](int r) {
    RankResult& out = res[(size_t)r];
    try {
      auto dev = GpuDevice::create(r);
      ArrowComputePipeline p(dev, "rank");
      Communicator comm(dev, id, r, world, 60000);  // collective: all threads arrive here, or give up after 60 s
      {  // what the line may claim: RCCL's own rank count and one identity record per rank gathered through the communicator
        auto pr = comm.peers(p);
        out.rccl_ranks = comm.size();
        out.distinct_devices = pr.second;
        if (out.rccl_ranks != world || pr.second != world)
          throw ArrowErrorGPU(ArrowErrorGPU::Runtime, "the communicator is not `world` distinct devices: ncclCommCount " +
                                                          std::to_string(out.rccl_ranks) + ", distinct devices " + std::to_string(pr.second));
      }
      const Shard sh = shard_rows(total, world, r);
      const uint64_t n = sh.rows, nb = agpu_bitmap_bytes(n);
      out.rows = n;
      // the bench step's nine buffers as two placed tables (agpu_malloc_table), like bench.py
      uint64_t sz3[3] = {4 * n, 4 * n, 4 * n}, sz6[6] = {4 * n, 4 * n, nb, nb, nb, nb};
      void *f[3], *c[6];
      check(agpu_malloc_table(dev->raw, 3, sz3, 0, f), "agpu_malloc_table");
      check(agpu_malloc_table(dev->raw, 6, sz6, 0, c), "agpu_malloc_table");
      check(agpu_synth_f32(p.h(), (float*)f[0], n, 20250418, sh.row0, -1000.0f, 1000.0f), "synth");
      check(agpu_synth_f32(p.h(), (float*)f[1], n, 20250419, sh.row0, -1000.0f, 1000.0f), "synth");
      check(agpu_synth_i32(p.h(), (int32_t*)c[0], n, 20250420, sh.row0, 1024), "synth");
      check(agpu_synth_i32(p.h(), (int32_t*)c[1], n, 20250421, sh.row0, 1024), "synth");
      check(agpu_synth_bits(p.h(), c[2], n, 20250422, sh.row0, 0.9), "synth");
      check(agpu_synth_bits(p.h(), c[3], n, 20250423, sh.row0, 0.9), "synth");
      auto step = [&](agpu_event** ev) {
        if (ev) check(agpu_event_record(ev[0], p.h()), "event");
        check(agpu_binary(p.h(), AGPU_OP_ADD, AGPU_F32, f[0], f[1], f[2], n), "agpu_binary");
        if (ev) check(agpu_event_record(ev[1], p.h()), "event");
        check(agpu_compare_validity(p.h(), AGPU_CMP_EQ, AGPU_I32, c[0], c[1], c[2], c[3], c[4], c[5], n), "agpu_compare_validity");
        if (ev) check(agpu_event_record(ev[2], p.h()), "event");
      };
      std::vector<std::array<agpu_event*, 3>> evs((size_t)steps);
      for (auto& e : evs)
        for (auto& x : e) check(agpu_event_create(dev->raw, &x), "agpu_event_create");
      for (int w = 0; w < 3; w++) step(nullptr);
      p.sync();
      comm.barrier(p);
      auto t0 = std::chrono::steady_clock::now();
      for (int s = 0; s < steps; s++) step(evs[(size_t)s].data());
      p.sync();
      comm.barrier(p);
      out.step_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      for (auto& e : evs) {
        float a = 0, b = 0;
        check(agpu_event_elapsed_ms(e[0], e[1], &a), "elapsed");
        check(agpu_event_elapsed_ms(e[1], e[2], &b), "elapsed");
        out.add_ms += a / steps;
        out.eq_ms += b / steps;
        for (auto& x : e) agpu_event_destroy(x);
      }
      // whole-column statistics of the first f32 column with the RCCL final reduce
      {
        auto buf = std::make_shared<Buffer>();
        buf->ptr = f[0];
        buf->bytes = 4 * n;
        buf->dev = dev;  // the Buffer frees f[0] (an ordinary member of the table block) when the array goes
        Float32ArrayGPU shard(buf, dev, n, std::nullopt);
        (void)sum_sharded_op(shard, comm, p);  // warm-up: scratch + RCCL's first call
        comm.sync(p);
        comm.barrier(p);
        t0 = std::chrono::steady_clock::now();
        auto s = sum_sharded_op(shard, comm, p);
        auto lo = min_sharded_op(shard, comm, p);
        auto hi = max_sharded_op(shard, comm, p);
        comm.sync(p);  // the wait with the collective deadline (p.sync() would block for ever behind a collective a dead peer never joins)
        comm.barrier(p);
        out.stats_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        out.sum = s.raw_values()[0];
        out.mn = lo.raw_values()[0];
        out.mx = hi.raw_values()[0];
        // … and the same statistics (+ the f64-accumulated sum) from ONE read of the shard: every field must be what the calls above gave
        (void)stats_sharded_op(shard, comm, p);
        comm.sync(p);
        comm.barrier(p);
        t0 = std::chrono::steady_clock::now();
        auto one = stats_sharded_op(shard, comm, p);
        comm.sync(p);
        comm.barrier(p);
        out.one_pass_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const agpu_f32_stats st = one.values();
        out.one_pass_same = memcmp(&st.sum, &out.sum, 4) == 0 && memcmp(&st.min, &out.mn, 4) == 0 && memcmp(&st.max, &out.mx, 4) == 0;
      }
      for (int k = 1; k < 3; k++) agpu_free(dev->raw, f[k]);
      for (int k = 0; k < 6; k++) agpu_free(dev->raw, c[k]);
    } catch (const std::exception& e) {
      out.error = e.what();
    }
  };
  std::vector<std::thread> ts;
  for (int r = 1; r < world; r++) ts.emplace_back(rank_main, r);
  rank_main(0);
  for (auto& t : ts) t.join();
  bool ok = true;
  double stats_ms = 0, step_s = 0, add_lo = 1e30, add_hi = 0, eq_lo = 1e30, eq_hi = 0, one_ms = 0;
  bool one_same = true;
  for (int r = 0; r < world; r++) {
    const RankResult& x = res[(size_t)r];
    if (!x.error.empty()) {
      fprintf(stderr, "rank %d failed: %s\n", r, x.error.c_str());
      ok = false;
    }
    ok = ok && memcmp(&x.sum, &res[0].sum, 4) == 0 && x.mn == res[0].mn && x.mx == res[0].mx;
    stats_ms = std::max(stats_ms, x.stats_ms);
    one_ms = std::max(one_ms, x.one_pass_ms);
    one_same = one_same && x.one_pass_same;
    step_s = std::max(step_s, x.step_s);
    add_lo = std::min(add_lo, x.add_ms), add_hi = std::max(add_hi, x.add_ms);
    eq_lo = std::min(eq_lo, x.eq_ms), eq_hi = std::max(eq_hi, x.eq_ms);
  }
  if (!ok) {
    // a rank that failed may have left its peers' rendezvous pending on a helper thread: end the process
    fflush(stderr);
    _Exit(1);
  }
  printf("{\"what\": \"C++ host, one thread per GPU: f32 add + i32 eq with validity (bench.py's step), then chunk-sharded f32 "
         "sum/min/max with the RCCL final reduce\", \"world\": %d, \"rccl_ranks\": %d, \"distinct_devices\": %d, \"scaling\": \"%s\", \"rows_total\": %llu, \"rows_rank0\": %llu, "
         "\"steps\": %d, \"value_GBps\": %.2f, \"ms_per_step\": %.4f, \"add_ms\": {\"min\": %.4f, \"max\": %.4f}, "
         "\"eq_ms\": {\"min\": %.4f, \"max\": %.4f}, \"sum\": %.9g, \"min\": %.9g, \"max\": %.9g, \"ms_3_statistics\": %.4f, "
         "\"statistics_aggregate_GBps\": %.1f, \"ms_one_pass_statistics\": %.4f, \"one_pass_identical\": %s, \"identical_on_all_ranks\": true, \"runtime\": \"%s\"}\n",
         world, res[0].rccl_ranks, res[0].distinct_devices, strong ? "strong" : "weak", (unsigned long long)total, (unsigned long long)res[0].rows, steps,
         20.5 * (double)total * steps / step_s / 1e9, step_s / steps * 1e3, add_lo, add_hi, eq_lo, eq_hi, res[0].sum, res[0].mn,
         res[0].mx, stats_ms, 3.0 * 4.0 * (double)total / stats_ms / 1e6, one_ms, one_same ? "true" : "false", Communicator::runtime_info().c_str());
  fflush(stdout);  // before the static destructors of the runtimes underneath
  return 0;
}
