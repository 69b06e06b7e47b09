// libclray_hip.so - host side of the MI355X back end (see include/clray_hip.h).
// Compiles scene-specialised HIP source with hiprtc for gfx950, loads it as a module and
// launches its kernels.  The only device-side object in this file is the static addend log of
// clr_render_passes_static_log; the kernels are the scene-specialised programs.
#include "clray_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_error = "";

int fail(const std::string &what) {
    g_error = what;
    return 1;
}

int check(hipError_t e, const char *what) {
    if (e == hipSuccess) return 0;
    return fail(std::string(what) + ": " + hipGetErrorString(e));
}

#define CLR_TRY(expr) do { if (int rc_ = check((expr), #expr)) return rc_; } while (0)

const unsigned kBlock = 256;
// development knob: threads per block of the render kernel (64, 128 or 256 = default); the kernels
// index with blockDim, so results do not depend on it
unsigned render_block() {
    static const unsigned value = [] {
        const char *e = getenv("CLR_RENDER_BLOCK");
        const unsigned v = e ? (unsigned)atoi(e) : 0;
        return (v == 64 || v == 128) ? v : kBlock;
    }();
    return value;
}

// ---- addend log of clr_render_passes_static_log: device memory the library holds from load to
// exit so that the call itself allocates nothing.  On the host `scratch` is only a handle for
// hipGetSymbolAddress; kernels get the per-device address kept in g_scratch_ptr.
static __device__ __align__(256) unsigned char scratch[CLR_STATIC_LOG_BYTES];
const int kMaxDevices = 64;
void *g_scratch_ptr[kMaxDevices];
// the last call that used a device's log: its stream and an event behind its last kernel, so that a
// call on ANOTHER stream that would overlap it is refused instead of corrupting the log
hipEvent_t g_scratch_event[kMaxDevices];
void *g_scratch_stream[kMaxDevices];
bool g_scratch_used[kMaxDevices];
std::mutex g_scratch_mutex;

// with `device` current: looks up that device's copy of the array once (and makes the event: the
// entry point itself creates nothing)
int resolve_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return fail("device index beyond the static log table");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return 0;
    void *ptr = nullptr;
    if (int rc = check(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(static log)")) return rc;
    if (!ptr) return fail("hipGetSymbolAddress(static log) returned a null address");
    if (int rc = check(hipEventCreateWithFlags(&g_scratch_event[device], hipEventDisableTiming), "hipEventCreate(static log)")) return rc;
    g_scratch_ptr[device] = ptr;
    return 0;
}

}  // namespace

// ---- stable compaction of the non-negative elements (clr_find_non_negative)
// A tile = 16 rounds of 256 consecutive elements; within a round the position of an element among
// the kept ones is its rank in its wave's ballot plus the kept counts of the waves before it.
namespace {

constexpr int kCompactBlock = 256, kCompactRounds = 16, kCompactTile = kCompactBlock * kCompactRounds;
constexpr int kCompactWaves = kCompactBlock / 64;

__global__ void __launch_bounds__(kCompactBlock)
compact_count(const int * __restrict__ in, const size_t n, unsigned * __restrict__ tile_count) {
    __shared__ unsigned wave_total[kCompactWaves];
    const size_t base = (size_t)blockIdx.x * kCompactTile;
    unsigned kept = 0;
    for (int r = 0; r < kCompactRounds; ++r) {
        const size_t i = base + (size_t)r * kCompactBlock + threadIdx.x;
        kept += __popcll(__ballot(i < n && in[i] >= 0));      // the same number in every lane of the wave
    }
    if ((threadIdx.x & 63) == 0) wave_total[threadIdx.x >> 6] = kept;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned sum = 0;
        for (int w = 0; w < kCompactWaves; ++w) sum += wave_total[w];
        tile_count[blockIdx.x] = sum;
    }
}

// one block: counts -> exclusive offsets in place, total behind them
__global__ void __launch_bounds__(1024)
compact_scan(unsigned * __restrict__ tile_count, const size_t n_tiles) {
    __shared__ unsigned s[1024];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (size_t first = 0; first < n_tiles; first += 1024) {
        const size_t i = first + threadIdx.x;
        const unsigned own = i < n_tiles ? tile_count[i] : 0;
        s[threadIdx.x] = own;
        __syncthreads();
        for (int step = 1; step < 1024; step <<= 1) {
            const unsigned add = (int)threadIdx.x >= step ? s[threadIdx.x - step] : 0;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_tiles) tile_count[i] = carry + s[threadIdx.x] - own;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_count[n_tiles] = carry;
}

__global__ void __launch_bounds__(kCompactBlock)
compact_scatter(const int * __restrict__ in, int * __restrict__ out, const size_t n,
                const unsigned * __restrict__ tile_offset) {
    __shared__ unsigned wave_total[2][kCompactWaves];
    const size_t base = (size_t)blockIdx.x * kCompactTile;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    size_t running = tile_offset[blockIdx.x];
    for (int r = 0; r < kCompactRounds; ++r) {
        const size_t i = base + (size_t)r * kCompactBlock + threadIdx.x;
        const int value = i < n ? in[i] : -1;
        const bool keep = value >= 0;
        const unsigned long long votes = __ballot(keep);
        if (lane == 0) wave_total[r & 1][wave] = __popcll(votes);
        __syncthreads();                    // one barrier a round: the two halves of wave_total alternate
        unsigned before = 0, all = 0;
        for (int w = 0; w < kCompactWaves; ++w) {
            const unsigned t = wave_total[r & 1][w];
            before += w < wave ? t : 0;
            all += t;
        }
        if (keep) out[running + before + __popcll(votes & ((1ull << lane) - 1ull))] = value;
        running += all;
    }
}

}  // namespace

struct clr_ctx {
    int device;
    hipEvent_t ev_start, ev_stop;         // around one timed launch
    hipEvent_t ev_mid = nullptr;          // timed pass-split call: between the logging and the adding kernel
    float last_render_ms = 0.0f, last_reduce_ms = 0.0f;
    hipEvent_t timer_start, timer_stop;   // clr_timer_begin / clr_timer_end
    unsigned *compact_work;               // clr_find_non_negative: per-tile counts / offsets, then the total
    size_t compact_tiles;                 // tiles the work buffer has room for
};

struct clr_program {
    clr_ctx *ctx;
    hipModule_t module;
    hipFunction_t render, render_debug, first_hit, reduce_log;
    int rng_mode;      // 1: the program draws per-pixel random numbers, 0: it reads the bounce tape
    int split_mode;    // 1: clr_render logs addends per pass and clr_reduce_log sums them
    int log_floats;    // floats per (pass, pixel) of the addend log (module global clr_log_floats)
    int regen_mode;    // 1: regenerating form of clr_render, 0: lock-step form, -1: not stated
};

namespace {

int compile(const char *source, int n_options, const char **options, const std::string &arch,
            std::vector<char> &code) {
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, source, "clray_scene.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail("hiprtcCreateProgram failed");
    std::vector<std::string> opts;
    opts.push_back("--offload-arch=" + arch);
    for (int i = 0; i < n_options; ++i) opts.push_back(options[i]);
    std::vector<const char *> raw;
    for (auto &o : opts) raw.push_back(o.c_str());
    hiprtcResult r = hiprtcCompileProgram(prog, (int)raw.size(), raw.data());
    size_t log_size = 0;
    hiprtcGetProgramLogSize(prog, &log_size);
    std::string log(log_size, '\0');
    if (log_size) hiprtcGetProgramLog(prog, &log[0]);
    if (r != HIPRTC_SUCCESS) {
        hiprtcDestroyProgram(&prog);
        return fail(std::string("hiprtc: ") + hiprtcGetErrorString(r) + "\n" + log);
    }
    size_t size = 0;
    hiprtcGetCodeSize(prog, &size);
    code.resize(size);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return 0;
}

int set_device(clr_ctx *ctx) {
    if (!ctx) return fail("null context");
    return check(hipSetDevice(ctx->device), "hipSetDevice");
}

int launch(clr_ctx *ctx, hipFunction_t fn, const unsigned grid[3], const unsigned block[3],
           void **args, void *stream, float *elapsed_ms) {
    hipStream_t s = (hipStream_t)stream;
    if (elapsed_ms) CLR_TRY(hipEventRecord(ctx->ev_start, s));
    CLR_TRY(hipModuleLaunchKernel(fn, grid[0], grid[1], grid[2], block[0], block[1], block[2],
                                  0, s, args, nullptr));
    if (elapsed_ms) {
        CLR_TRY(hipEventRecord(ctx->ev_stop, s));
        CLR_TRY(hipEventSynchronize(ctx->ev_stop));
        CLR_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev_start, ctx->ev_stop));
    }
    return 0;
}

}  // namespace

// lets the other translation units of the library report through clr_last_error
extern "C" __attribute__((visibility("hidden"))) void clr_set_error(const char *text) { g_error = text; }
extern "C" __attribute__((visibility("hidden"))) int clr_ctx_device(const clr_ctx *ctx) { return ctx->device; }

extern "C" {

const char *clr_last_error(void) { return g_error.c_str(); }

int clr_ctx_create(int device_index, clr_ctx **out) {
    if (!out) return fail("null output pointer");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail("no HIP device available: libclray_hip needs an AMD GPU (gfx950)");
    if (device_index < 0 || device_index >= count) return fail("device index out of range");
    CLR_TRY(hipSetDevice(device_index));
    // the static log of clr_render_passes_static_log is looked up here, at set-up; a device for
    // which that fails (no code object of the library for it) can still run everything else, and
    // the entry point reports the failure when it is called
    if (resolve_scratch(device_index) != 0) (void)hipGetLastError();
    clr_ctx *ctx = new clr_ctx;
    ctx->device = device_index;
    ctx->compact_work = nullptr;
    ctx->compact_tiles = 0;
    if (int rc = check(hipEventCreate(&ctx->ev_start), "hipEventCreate")) { delete ctx; return rc; }
    if (int rc = check(hipEventCreate(&ctx->ev_stop), "hipEventCreate")) { delete ctx; return rc; }
    if (int rc = check(hipEventCreate(&ctx->ev_mid), "hipEventCreate")) { delete ctx; return rc; }
    if (int rc = check(hipEventCreate(&ctx->timer_start), "hipEventCreate")) { delete ctx; return rc; }
    if (int rc = check(hipEventCreate(&ctx->timer_stop), "hipEventCreate")) { delete ctx; return rc; }
    *out = ctx;
    return 0;
}

int clr_ctx_destroy(clr_ctx *ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipEventDestroy(ctx->ev_start);
    (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->ev_mid) (void)hipEventDestroy(ctx->ev_mid);
    (void)hipEventDestroy(ctx->timer_start);
    (void)hipEventDestroy(ctx->timer_stop);
    if (ctx->compact_work) (void)hipFree(ctx->compact_work);
    delete ctx;
    return 0;
}

int clr_device_info(clr_ctx *ctx, clr_device_info_t *out) {
    if (int rc = set_device(ctx)) return rc;
    if (!out) return fail("null output pointer");
    hipDeviceProp_t p;
    CLR_TRY(hipGetDeviceProperties(&p, ctx->device));
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", p.name);
    snprintf(out->arch, sizeof(out->arch), "%s", p.gcnArchName);
    out->compute_units = p.multiProcessorCount;
    out->clock_khz = p.clockRate;
    out->wavefront_size = p.warpSize;
    out->lds_bytes_per_block = (int)p.sharedMemPerBlock;
    out->global_mem_bytes = p.totalGlobalMem;
    return 0;
}

int clr_sync(clr_ctx *ctx, void *stream) {
    if (int rc = set_device(ctx)) return rc;
    CLR_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int clr_stream_create(clr_ctx *ctx, void **stream) {
    if (int rc = set_device(ctx)) return rc;
    if (!stream) return fail("null output pointer");
    hipStream_t s = nullptr;
    CLR_TRY(hipStreamCreate(&s));
    *stream = (void *)s;
    return 0;
}

int clr_stream_destroy(clr_ctx *ctx, void *stream) {
    if (int rc = set_device(ctx)) return rc;
    if (!stream) return 0;
    CLR_TRY(hipStreamSynchronize((hipStream_t)stream));
    CLR_TRY(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

int clr_stream_wait(clr_ctx *ctx, void *waiting, void *on) {
    if (int rc = set_device(ctx)) return rc;
    if (waiting == on) return 0;
    hipEvent_t reached;
    CLR_TRY(hipEventCreateWithFlags(&reached, hipEventDisableTiming));
    hipError_t e = hipEventRecord(reached, (hipStream_t)on);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiting, reached, 0);
    (void)hipEventDestroy(reached);      // released once the recorded work has completed
    return check(e, "clr_stream_wait");
}

int clr_timer_begin(clr_ctx *ctx, void *stream) {
    if (int rc = set_device(ctx)) return rc;
    CLR_TRY(hipEventRecord(ctx->timer_start, (hipStream_t)stream));
    return 0;
}

int clr_timer_end(clr_ctx *ctx, void *stream) {
    if (int rc = set_device(ctx)) return rc;
    CLR_TRY(hipEventRecord(ctx->timer_stop, (hipStream_t)stream));
    return 0;
}

int clr_timer_elapsed(clr_ctx *ctx, float *elapsed_ms) {
    if (int rc = set_device(ctx)) return rc;
    if (!elapsed_ms) return fail("null output pointer");
    CLR_TRY(hipEventSynchronize(ctx->timer_stop));
    CLR_TRY(hipEventElapsedTime(elapsed_ms, ctx->timer_start, ctx->timer_stop));
    return 0;
}

int clr_compile_only(const char *source, int n_options, const char **options, const char *arch,
                     size_t *code_size) {
    if (!source) return fail("null source");
    std::vector<char> code;
    if (int rc = compile(source, n_options, options, arch ? arch : "gfx950", code)) return rc;
    if (code_size) *code_size = code.size();
    return 0;
}

int clr_compile(const char *source, int n_options, const char **options, const char *arch,
                void **binary, size_t *size) {
    if (!source || !binary || !size) return fail("null argument");
    std::vector<char> code;
    if (int rc = compile(source, n_options, options, arch ? arch : "gfx950", code)) return rc;
    void *copy = malloc(code.size() ? code.size() : 1);
    if (!copy) return fail("out of memory");
    memcpy(copy, code.data(), code.size());
    *binary = copy;
    *size = code.size();
    return 0;
}

void clr_binary_free(void *binary) { free(binary); }

int clr_compiler_version(int *major, int *minor) {
    int a = 0, b = 0;
    if (hiprtcVersion(&a, &b) != HIPRTC_SUCCESS) return fail("hiprtcVersion failed");
    if (major) *major = a;
    if (minor) *minor = b;
    return 0;
}

namespace {
int load_program(clr_ctx *ctx, const void *binary, clr_program **out);
}

int clr_program_build(clr_ctx *ctx, const char *source, int n_options, const char **options,
                      clr_program **out) {
    if (int rc = set_device(ctx)) return rc;
    if (!source || !out) return fail("null argument");
    hipDeviceProp_t p;
    CLR_TRY(hipGetDeviceProperties(&p, ctx->device));
    std::vector<char> code;
    if (int rc = compile(source, n_options, options, p.gcnArchName, code)) return rc;
    return load_program(ctx, code.data(), out);
}

int clr_program_load(clr_ctx *ctx, const void *binary, size_t size, clr_program **out) {
    if (int rc = set_device(ctx)) return rc;
    if (!binary || !out) return fail("null argument");
    if (size < 4 || memcmp(binary, "\177ELF", 4) != 0)
        if (size < 24 || memcmp(binary, "__CLANG_OFFLOAD_BUNDLE__", 24) != 0)
            return fail("clr_program_load: not a code object");
    return load_program(ctx, binary, out);
}

namespace {
int load_program(clr_ctx *ctx, const void *binary, clr_program **out) {
    clr_program *prog = new clr_program;
    prog->ctx = ctx;
    if (int rc = check(hipModuleLoadData(&prog->module, binary), "hipModuleLoadData")) {
        delete prog;
        return rc;
    }
    // a program for clr_launch only (unit kernels) need not hold the two fixed entry kernels
    if (hipModuleGetFunction(&prog->render, prog->module, "clr_render") != hipSuccess) prog->render = nullptr;
    if (hipModuleGetFunction(&prog->render_debug, prog->module, "clr_render_debug") != hipSuccess) prog->render_debug = nullptr;
    if (hipModuleGetFunction(&prog->first_hit, prog->module, "clr_first_hit") != hipSuccess) prog->first_hit = nullptr;
    if (hipModuleGetFunction(&prog->reduce_log, prog->module, "clr_reduce_log") != hipSuccess) prog->reduce_log = nullptr;
    prog->rng_mode = 0;
    prog->split_mode = 0;
    hipDeviceptr_t flag;
    size_t flag_size = 0;
    if (hipModuleGetGlobal(&flag, &flag_size, prog->module, "clr_rng_mode") == hipSuccess && flag_size == sizeof(int))
        (void)hipMemcpyDtoH(&prog->rng_mode, flag, sizeof(int));
    if (hipModuleGetGlobal(&flag, &flag_size, prog->module, "clr_split_mode") == hipSuccess && flag_size == sizeof(int))
        (void)hipMemcpyDtoH(&prog->split_mode, flag, sizeof(int));
    prog->regen_mode = -1;
    if (hipModuleGetGlobal(&flag, &flag_size, prog->module, "clr_regen_mode") == hipSuccess && flag_size == sizeof(int))
        (void)hipMemcpyDtoH(&prog->regen_mode, flag, sizeof(int));
    prog->log_floats = 0;
    if (hipModuleGetGlobal(&flag, &flag_size, prog->module, "clr_log_floats") == hipSuccess && flag_size == sizeof(int))
        (void)hipMemcpyDtoH(&prog->log_floats, flag, sizeof(int));
    (void)hipGetLastError();
    *out = prog;
    return 0;
}
}  // namespace

int clr_program_destroy(clr_program *prog) {
    if (!prog) return 0;
    (void)hipSetDevice(prog->ctx->device);
    (void)hipModuleUnload(prog->module);
    delete prog;
    return 0;
}

int clr_program_set_global(clr_program *prog, const char *name, const void *host, size_t bytes) {
    if (!prog || !name || !host) return fail("null argument");
    if (int rc = set_device(prog->ctx)) return rc;
    hipDeviceptr_t dptr;
    size_t size = 0;
    CLR_TRY(hipModuleGetGlobal(&dptr, &size, prog->module, name));
    if (bytes > size)
        return fail(std::string("clr_program_set_global: ") + name + " holds " + std::to_string(size) +
                    " bytes, asked to write " + std::to_string(bytes));
    CLR_TRY(hipMemcpyHtoD(dptr, const_cast<void *>(host), bytes));
    return 0;
}

int clr_program_get_global(clr_program *prog, const char *name, void *host, size_t bytes) {
    if (!prog || !name || !host) return fail("null argument");
    if (int rc = set_device(prog->ctx)) return rc;
    hipDeviceptr_t dptr;
    size_t size = 0;
    CLR_TRY(hipModuleGetGlobal(&dptr, &size, prog->module, name));
    if (bytes > size)
        return fail(std::string("clr_program_get_global: ") + name + " holds " + std::to_string(size) +
                    " bytes, asked to read " + std::to_string(bytes));
    CLR_TRY(hipDeviceSynchronize());
    CLR_TRY(hipMemcpyDtoH(host, dptr, bytes));
    return 0;
}

int clr_kernel_info(clr_program *prog, const char *kernel, clr_kernel_info_t *out) {
    if (!prog || !kernel || !out) return fail("null argument");
    if (int rc = set_device(prog->ctx)) return rc;
    hipFunction_t fn;
    CLR_TRY(hipModuleGetFunction(&fn, prog->module, kernel));
    memset(out, 0, sizeof(*out));
    CLR_TRY(hipFuncGetAttribute(&out->vgprs, HIP_FUNC_ATTRIBUTE_NUM_REGS, fn));
    CLR_TRY(hipFuncGetAttribute(&out->scratch_bytes, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fn));
    CLR_TRY(hipFuncGetAttribute(&out->lds_bytes, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, fn));
    CLR_TRY(hipFuncGetAttribute(&out->max_threads_per_block, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, fn));
    return 0;
}

int clr_buffer_alloc(clr_ctx *ctx, size_t bytes, void **device_ptr) {
    if (int rc = set_device(ctx)) return rc;
    if (!device_ptr) return fail("null output pointer");
    CLR_TRY(hipMalloc(device_ptr, bytes ? bytes : 4));
    return 0;
}

int clr_buffer_free(clr_ctx *ctx, void *device_ptr) {
    if (int rc = set_device(ctx)) return rc;
    CLR_TRY(hipFree(device_ptr));
    return 0;
}

int clr_buffer_upload(clr_ctx *ctx, void *device_ptr, const void *host, size_t bytes, void *stream) {
    if (int rc = set_device(ctx)) return rc;
    CLR_TRY(hipMemcpyAsync(device_ptr, host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    CLR_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int clr_buffer_download(clr_ctx *ctx, void *host, const void *device_ptr, size_t bytes, void *stream) {
    if (int rc = set_device(ctx)) return rc;
    CLR_TRY(hipMemcpyAsync(host, device_ptr, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    CLR_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int clr_buffer_fill32(clr_ctx *ctx, void *device_ptr, uint32_t value, size_t count, void *stream) {
    if (int rc = set_device(ctx)) return rc;
    CLR_TRY(hipMemsetD32Async((hipDeviceptr_t)device_ptr, (int)value, count, (hipStream_t)stream));
    return 0;
}

int clr_buffer_copy(clr_ctx *ctx, void *dest, const void *src, size_t bytes, void *stream) {
    if (int rc = set_device(ctx)) return rc;
    if (!bytes) return 0;
    if (!dest || !src) return fail("clr_buffer_copy: null buffer");
    CLR_TRY(hipMemcpyDtoDAsync((hipDeviceptr_t)dest, (hipDeviceptr_t)src, bytes, (hipStream_t)stream));
    return 0;
}

int clr_find_non_negative(clr_ctx *ctx, const int32_t *in, int32_t *out, size_t n, int *count, void *stream) {
    if (int rc = set_device(ctx)) return rc;
    if (!count) return fail("clr_find_non_negative: null count pointer");
    *count = 0;
    if (n == 0) return 0;
    if (!in || !out) return fail("clr_find_non_negative: null buffer");
    if (n > (size_t)0x7fffffff) return fail("clr_find_non_negative: more than 2^31 - 1 elements");
    const size_t n_tiles = (n + kCompactTile - 1) / kCompactTile;
    if (ctx->compact_tiles < n_tiles) {
        if (ctx->compact_work) CLR_TRY(hipFree(ctx->compact_work));
        ctx->compact_work = nullptr;
        ctx->compact_tiles = 0;
        CLR_TRY(hipMalloc((void **)&ctx->compact_work, (n_tiles + 1) * sizeof(unsigned)));
        ctx->compact_tiles = n_tiles;
    }
    hipStream_t s = (hipStream_t)stream;
    compact_count<<<dim3((unsigned)n_tiles), dim3(kCompactBlock), 0, s>>>(in, n, ctx->compact_work);
    compact_scan<<<dim3(1), dim3(1024), 0, s>>>(ctx->compact_work, n_tiles);
    compact_scatter<<<dim3((unsigned)n_tiles), dim3(kCompactBlock), 0, s>>>(in, out, n, ctx->compact_work);
    CLR_TRY(hipGetLastError());
    unsigned total = 0;
    CLR_TRY(hipMemcpyAsync(&total, ctx->compact_work + n_tiles, sizeof(total), hipMemcpyDeviceToHost, s));
    CLR_TRY(hipStreamSynchronize(s));
    *count = (int)total;
    return 0;
}

int clr_launch(clr_program *prog, const char *kernel, const unsigned grid[3],
               const unsigned block[3], int n_args, void **args, void *stream, float *elapsed_ms) {
    if (!prog || !kernel) return fail("null argument");
    (void)n_args;
    if (int rc = set_device(prog->ctx)) return rc;
    hipFunction_t fn;
    CLR_TRY(hipModuleGetFunction(&fn, prog->module, kernel));
    return launch(prog->ctx, fn, grid, block, args, stream, elapsed_ms);
}

namespace {
// static_log: the addend log is the library's own array (clr_render_passes_static_log)
int render_passes(clr_program *prog, const clr_render_args *a, void *stream, float *elapsed_ms,
                  void *static_log) {
    if (!prog || !a) return fail("null argument");
    if (a->struct_size != sizeof(clr_render_args))
        return fail("clr_render_passes: args->struct_size is " + std::to_string(a->struct_size) +
                    ", this library's clr_render_args has " + std::to_string(sizeof(clr_render_args)) +
                    " bytes: the binding was made from another version of clray_hip.h");
    if (int rc = set_device(prog->ctx)) return rc;
    if (!prog->render) return fail("clr_render_passes: the program has no kernel clr_render");
    if (a->n_pixels <= 0 || a->n_passes < 0) return fail("clr_render_passes: bad n_pixels / n_passes");
    if (!a->img || !a->cam || !a->pass_tape)
        return fail("clr_render_passes: img, cam and pass_tape are required");
    if ((a->per_pixel_rng != 0) != (prog->rng_mode != 0))
        return fail("clr_render_passes: per_pixel_rng does not match the random number mode the program was generated for");
    if (!prog->rng_mode && !a->bounce_tape)
        return fail("clr_render_passes: bounce_tape is required for a tape program");
    if (a->pass_stride < 0) return fail("clr_render_passes: pass_stride must not be negative");
    if (a->ray_band < 0 || (a->ray_band > 0 && (a->ray_band_stride < a->ray_band || a->ray_band_base < 0)))
        return fail("clr_render_passes: ray_band must be 0 or positive with ray_band_stride >= ray_band and ray_band_base >= 0");
    if (prog->regen_mode >= 0 && (a->regenerate != 0) != (prog->regen_mode != 0))
        return fail("clr_render_passes: `regenerate` does not match the form of clr_render the program was generated in");
    if (a->debug_dump && a->n_passes != 1)
        return fail("clr_render_passes: debug_dump needs n_passes == 1");
    if (a->debug_dump && !prog->render_debug)
        return fail("clr_render_passes: debug_dump given, but the program has no kernel clr_render_debug");
    // the dump is written by a kernel of its own; clr_render carries no code for it
    hipFunction_t render = a->debug_dump ? prog->render_debug : prog->render;
    if (static_log) {
        if (!prog->split_mode)
            return fail("clr_render_passes_static_log: only pass-split programs use a log; call clr_render_passes");
        if (a->pass_log) return fail("clr_render_passes_static_log: pass_log must be NULL, the log is the library's");
        if (!prog->reduce_log || prog->log_floats < 1)
            return fail("clr_render_passes_static_log: the program lacks clr_reduce_log / clr_log_floats");
        if (a->pass_groups < 1) return fail("clr_render_passes_static_log: pass_groups must be >= 1");
        // by division: the product of three 32-bit factors can wrap 64 bits
        const unsigned long long per_pass = 4ull * (unsigned)prog->log_floats * (unsigned)a->n_pixels;
        const unsigned long long need = per_pass * (unsigned)a->n_passes;
        if ((unsigned)a->n_passes > CLR_STATIC_LOG_BYTES / per_pass)
            return fail("clr_render_passes_static_log: the addend log of this call needs " + std::to_string(need) +
                        " bytes, the static log holds " + std::to_string(CLR_STATIC_LOG_BYTES) +
                        ": render fewer passes per call");
    }
    else if (prog->split_mode) {
        if (!prog->reduce_log) return fail("clr_render_passes: the pass-split program has no kernel clr_reduce_log");
        if (a->pass_groups < 1 || !a->pass_log)
            return fail("clr_render_passes: a pass-split program needs pass_groups >= 1 and a pass_log buffer");
    }
    else if (a->pass_groups != 0 || a->pass_log)
        return fail("clr_render_passes: pass_groups / pass_log given, but the program was not generated in pass-split mode");
    if (a->reduce_after && (!prog->split_mode || static_log))
        return fail("clr_render_passes: reduce_after is for pass-split programs with a pass_log of the caller's");
    if (a->n_passes == 0) { if (elapsed_ms) *elapsed_ms = 0.0f; return 0; }
    void *img = a->img;
    const void *cam = a->cam, *pt = a->pass_tape, *bt = a->bounce_tape, *mr = a->material_rows,
               *vd = a->vector_data, *id = a->integer_data;
    int n_pixels = a->n_pixels, n_passes = a->n_passes, regen = a->regenerate;
    void *dbg = a->debug_dump;
    unsigned seed_lo = a->rng_seed_lo, seed_hi = a->rng_seed_hi;
    int first_pass = a->first_pass, first_ray = a->first_ray;
    // every (pass, pixel) header of the log and every entry it counts is written by clr_render
    // before clr_reduce_log reads it, so what an earlier call left in the static array is never seen
    void *pass_log = static_log ? static_log : a->pass_log;
    int pass_groups = a->pass_groups;
    int pass_stride = a->pass_stride > 0 ? a->pass_stride : 1;
    int ray_band = a->ray_band, ray_band_stride = a->ray_band_stride, ray_band_base = a->ray_band_base;
    void *args[] = {&img, &cam, &pt, &bt, &mr, &vd, &id, &n_pixels, &n_passes, &regen, &dbg,
                    &seed_lo, &seed_hi, &first_pass, &first_ray, &pass_log, &pass_groups, &pass_stride,
                    &ray_band, &ray_band_stride, &ray_band_base};
    const unsigned rb = render_block();
    const unsigned block[3] = {rb, 1, 1};
    if (!prog->split_mode) {
        const unsigned grid[3] = {((unsigned)n_pixels + rb - 1) / rb, 1, 1};
        return launch(prog->ctx, render, grid, block, args, stream, elapsed_ms);
    }
    // pass-split program: n_pixels x pass_groups lanes write the log, then one lane per pixel sums it
    // whole waves per group, so that no wave straddles two groups (main.hip)
    const unsigned long long lanes = (((unsigned long long)n_pixels + 63) & ~63ull) * (unsigned)pass_groups;
    if (lanes > 0x7fffffffull) return fail("clr_render_passes: n_pixels x pass_groups is too large");
    const unsigned grid[3] = {(unsigned)((lanes + rb - 1) / rb), 1, 1};
    const unsigned reduce_grid[3] = {((unsigned)n_pixels + kBlock - 1) / kBlock, 1, 1};
    const void *log_in = pass_log;
    void *reduce_args[] = {&img, &log_in, &n_pixels, &n_passes};
    hipStream_t s = (hipStream_t)stream;
    if (elapsed_ms) CLR_TRY(hipEventRecord(prog->ctx->ev_start, s));
    CLR_TRY(hipModuleLaunchKernel(render, grid[0], 1, 1, rb, 1, 1, 0, s, args, nullptr));
    if (elapsed_ms) CLR_TRY(hipEventRecord(prog->ctx->ev_mid, s));
    // the sums of a pixel are added in call order: the adding kernel waits for the previous call's
    // on reduce_after, while the logging kernels of consecutive calls overlap (clray_hip.h)
    if (a->reduce_after && a->reduce_after != stream) {
        if (int rc = clr_stream_wait(prog->ctx, stream, a->reduce_after)) return rc;
    }
    CLR_TRY(hipModuleLaunchKernel(prog->reduce_log, reduce_grid[0], 1, 1, kBlock, 1, 1, 0, s, reduce_args, nullptr));
    if (elapsed_ms) {
        CLR_TRY(hipEventRecord(prog->ctx->ev_stop, s));
        CLR_TRY(hipEventSynchronize(prog->ctx->ev_stop));
        CLR_TRY(hipEventElapsedTime(elapsed_ms, prog->ctx->ev_start, prog->ctx->ev_stop));
        CLR_TRY(hipEventElapsedTime(&prog->ctx->last_render_ms, prog->ctx->ev_start, prog->ctx->ev_mid));
        CLR_TRY(hipEventElapsedTime(&prog->ctx->last_reduce_ms, prog->ctx->ev_mid, prog->ctx->ev_stop));
    }
    return 0;
}
}  // namespace

int clr_last_split_ms(clr_ctx *ctx, float *render_ms, float *reduce_ms) {
    if (!ctx) return fail("null argument");
    if (render_ms) *render_ms = ctx->last_render_ms;
    if (reduce_ms) *reduce_ms = ctx->last_reduce_ms;
    return 0;
}

int clr_render_passes(clr_program *prog, const clr_render_args *a, void *stream, float *elapsed_ms) {
    return render_passes(prog, a, stream, elapsed_ms, nullptr);
}

int clr_render_passes_static_log(clr_program *prog, const clr_render_args *a, void *stream, float *elapsed_ms) {
    if (!prog) return fail("null argument");
    const int device = prog->ctx->device;
    if (device < 0 || device >= kMaxDevices) return fail("clr_render_passes_static_log: device index beyond the static log table");
    if (int rc = set_device(prog->ctx)) return rc;
    if (!g_scratch_ptr[device])
        if (int rc = resolve_scratch(device)) return rc;          // set-up failed at clr_ctx_create: say why
    // one log per GPU: a call on another stream while the last one is still running would write into
    // the log that call is reading.  (Calls on one stream are ordered; a caller that orders its
    // streams itself synchronises before the next call.)
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_used[device] && g_scratch_stream[device] != stream &&
        hipEventQuery(g_scratch_event[device]) == hipErrorNotReady)
        return fail("clr_render_passes_static_log: the previous call on this GPU is still running on another "
                    "stream and owns the static log; wait for it (one log per GPU)");
    (void)hipGetLastError();
    const int rc = render_passes(prog, a, stream, elapsed_ms, g_scratch_ptr[device]);
    if (rc == 0 && a && a->n_passes > 0) {
        CLR_TRY(hipEventRecord(g_scratch_event[device], (hipStream_t)stream));
        g_scratch_stream[device] = stream;
        g_scratch_used[device] = true;
    }
    return rc;
}

int clr_first_hit(clr_program *prog, const void *cam, const void *pass_tape, const void *vector_data,
                  const void *integer_data, int n_pixels, void *out_ids, void *out_t,
                  void *out_normal, void *stream) {
    if (!prog) return fail("null argument");
    if (int rc = set_device(prog->ctx)) return rc;
    if (!prog->first_hit) return fail("clr_first_hit: the program has no kernel clr_first_hit");
    if (n_pixels <= 0 || !cam || !pass_tape || !out_ids || !out_t || !out_normal)
        return fail("clr_first_hit: bad arguments");
    void *args[] = {&cam, &pass_tape, &vector_data, &integer_data, &n_pixels, &out_ids, &out_t, &out_normal};
    const unsigned grid[3] = {((unsigned)n_pixels + kBlock - 1) / kBlock, 1, 1};
    const unsigned block[3] = {kBlock, 1, 1};
    return launch(prog->ctx, prog->first_hit, grid, block, args, stream, nullptr);
}

}  // extern "C"
