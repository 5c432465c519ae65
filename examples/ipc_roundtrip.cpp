// Arrow IPC file → HBM → kernels → Arrow IPC file from a compiled host (host/arrow_gpu.hpp IpcReader / IpcWriter over the C
// ABI's agpu_ipc_*).  The reference can only build arrays from host Vecs
// [ref: crates/array/src/array/primitive_array_gpu.rs:22-104]; a Rust host bound to this ABI reads its columns straight
// from the files arrow-rs / pyarrow write.
//
//   ipc_roundtrip --schema in.arrow            host-only: print the schema and the record batch sizes (no GPU needed)
//   ipc_roundtrip in.arrow out.arrow           per record batch: sum = a + b (f32, validity AND), eq = (k == m) (i32 → bool);
//                                              written as an Arrow IPC file with the columns "sum" and "eq"
// build: hipcc -std=c++17 -O2 -x c++ examples/ipc_roundtrip.cpp -Larrow_gpu_amd/lib -larrow_gpu_hip -Wl,-rpath,$PWD/arrow_gpu_amd/lib
#include <fcntl.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>

#include "../host/arrow_gpu.hpp"

using namespace arrow_gpu;

int main(int argc, char** argv) {
  try {
    if (argc == 3 && !strcmp(argv[1], "--schema")) {
      auto r = IpcReader::map_file(argv[2]);
      printf("{\"fields\": [");
      for (size_t i = 0; i < r->fields().size(); i++) {
        const IpcField& f = r->fields()[i];
        printf("%s{\"name\": \"%s\", \"format\": \"%s\", \"dtype\": %d, \"nullable\": %s}", i ? ", " : "", f.name.c_str(), f.format.c_str(),
               f.dtype, f.nullable ? "true" : "false");
      }
      printf("], \"batch_rows\": [");
      for (int64_t b = 0; b < r->num_batches(); b++) printf("%s%lld", b ? ", " : "", (long long)r->batch_rows(b));
      printf("]}\n");
      return 0;
    }
    if (argc != 3) {
      fprintf(stderr, "usage: %s --schema in.arrow | in.arrow out.arrow\n", argv[0]);
      return 1;
    }
    DevicePtr dev;
    try {
      dev = GpuDevice::create(0);
    } catch (const ArrowErrorGPU& e) {
      printf("no device: %s\n", e.what());  // there is no CPU fallback
      return 2;
    }
    auto r = IpcReader::map_file(argv[1]);
    const int ca = r->column_index("a"), cb = r->column_index("b"), ck = r->column_index("k"), cm = r->column_index("m");
    const int fd = ::open(argv[2], O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) {
      fprintf(stderr, "cannot create %s\n", argv[2]);
      return 1;
    }
    uint64_t bytes = 0;
    int64_t rows = 0;
    {
      IpcWriter w({IpcField{"sum", "", AGPU_F32, true}, IpcField{"eq", "", AGPU_BOOL, true}}, /*file_format=*/true, fd);
      for (int64_t b = 0; b < r->num_batches(); b++) {
        ArrowComputePipeline p(dev);
        auto cols = r->read_batch_op(b, {ca, cb, ck, cm}, p);  // one block, placed for the HBM channel hash
        auto a = try_from<Float32ArrayGPU>(cols[0]);
        auto bb = try_from<Float32ArrayGPU>(cols[1]);
        auto k = try_from<Int32ArrayGPU>(cols[2]);
        auto m = try_from<Int32ArrayGPU>(cols[3]);
        auto sum = a.add_op(bb, p);
        auto eq = k.eq_op(m, p);
        p.finish();
        w.write_batch({ArrowArrayGPU(sum), ArrowArrayGPU(eq)});
        rows += (int64_t)sum.len;
      }
      w.finish(&bytes);
    }
    ::close(fd);
    printf("{\"ok\": true, \"batches\": %lld, \"rows\": %lld, \"bytes_written\": %llu}\n", (long long)r->num_batches(), (long long)rows,
           (unsigned long long)bytes);
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
