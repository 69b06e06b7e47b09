/* clray_hip.h - C ABI of libclray_hip.so, the MI355X (gfx950) back end of the clray path tracer.
 *
 * This is the drop-in boundary for the reference's device seam, i.e. the methods of
 * `Accelerator` (/root/reference/accelerator.py) that `Renderer` (/root/reference/renderer.py)
 * calls, plus the per-pass / per-bounce launch schedule of `Renderer.render_sample`, which this
 * library runs as ONE kernel launch for a whole batch of sample passes.
 *
 * Conventions: plain C, no C++ or torch types; every function returns 0 on success and a
 * non-zero code on failure, and `clr_last_error()` then returns a message (for a failed
 * program build it includes the hiprtc log).  Handles are opaque pointers owned by the caller,
 * who must destroy them.  Device buffers are raw device addresses (`void *`), so memory
 * allocated elsewhere in the process (e.g. a torch tensor's data_ptr) can be passed in.
 * `stream` arguments are `hipStream_t` values cast to `void *`; NULL is the default stream.
 * One host thread per context; contexts of different GPUs are independent.
 */
#ifndef CLRAY_HIP_H
#define CLRAY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clr_ctx clr_ctx;
typedef struct clr_program clr_program;

typedef struct clr_device_info_t {
    char name[128];
    char arch[64];            /* e.g. "gfx950:sramecc+:xnack-" */
    int compute_units;
    int clock_khz;            /* peak shader clock */
    int wavefront_size;
    int lds_bytes_per_block;
    uint64_t global_mem_bytes;
} clr_device_info_t;

/* Message for the most recent failure on this thread (never NULL). */
const char *clr_last_error(void);

/* ---- context: replaces Accelerator.__init__ (accelerator.py:17-34) and finish (:36-41) */
int clr_ctx_create(int device_index, clr_ctx **out);
int clr_ctx_destroy(clr_ctx *ctx);
int clr_device_info(clr_ctx *ctx, clr_device_info_t *out);
int clr_sync(clr_ctx *ctx, void *stream);
/* Further in-order queues of the context's GPU (the reference has the one queue of
 * accelerator.py:28-29).  Launch chains that touch disjoint slices of the frame can go to streams of
 * their own: the blocks of one chain then fill the GPU while the last blocks of another drain
 * (DESIGN section 6).  The streams are ordinary (blocking) HIP streams: work on the default stream
 * (NULL) - copies, fills, clr_timer_* - waits for them and they wait for it.  clr_stream_wait
 * makes `waiting` (NULL = default stream) wait for everything queued on `on` so far, without
 * blocking the host. */
int clr_stream_create(clr_ctx *ctx, void **stream);
int clr_stream_destroy(clr_ctx *ctx, void *stream);
int clr_stream_wait(clr_ctx *ctx, void *waiting, void *on);

/* ---- program: replaces Accelerator.build_program (accelerator.py:43-49).  `source` is the
 * scene-specialised HIP translation unit; it is compiled with hiprtc for gfx950 (or for
 * `arch` when not NULL) and loaded as a module on the context's device. */
int clr_program_build(clr_ctx *ctx, const char *source, int n_options, const char **options,
                      clr_program **out);
int clr_program_destroy(clr_program *prog);
/* Copies `bytes` from host memory into the module-scope device variable `name`
 * (the __constant__ scene tables: parameters, data offsets, materials). */
int clr_program_set_global(clr_program *prog, const char *name, const void *host, size_t bytes);
/* The other direction: waits for the device and copies `bytes` of the module global `name` to
 * host memory (no counterpart in the reference; counters that a development build of a program
 * keeps, tools/mesh_phases.py). */
int clr_program_get_global(clr_program *prog, const char *name, void *host, size_t bytes);
/* Compile only (no device needed): returns the code-object size; used by the build check. */
int clr_compile_only(const char *source, int n_options, const char **options, const char *arch,
                     size_t *code_size);
/* The two halves of clr_program_build, for callers that keep compiled programs (a disk cache)
 * or compile in another process: compile for `arch` without a device (the binary is released
 * with clr_binary_free), and load a code object made earlier.  The reference rebuilds its
 * program from source on every start (accelerator.py:43-49; pyopencl keeps its own binary
 * cache underneath). */
int clr_compile(const char *source, int n_options, const char **options, const char *arch,
                void **binary, size_t *size);
void clr_binary_free(void *binary);
int clr_program_load(clr_ctx *ctx, const void *binary, size_t size, clr_program **out);
/* Version of the hiprtc that this process compiles with (part of a cache key). */
int clr_compiler_version(int *major, int *minor);

/* Resources of one kernel of a loaded program, as the code object states them (the reference
 * prints per-kernel statistics only, accelerator.py:155-166; this is what an occupancy choice
 * needs): registers per lane, bytes of scratch ("private segment", register spills and stack)
 * per lane, static LDS bytes per block. */
typedef struct clr_kernel_info_t {
    int vgprs;
    int scratch_bytes;
    int lds_bytes;
    int max_threads_per_block;
} clr_kernel_info_t;
int clr_kernel_info(clr_program *prog, const char *kernel, clr_kernel_info_t *out);

/* ---- buffers: replace to_device / new_array / new_const_buffer / enqueue_copy / Array.fill /
 * Array.get (accelerator.py:91-153, renderer.py:306-311) */
int clr_buffer_alloc(clr_ctx *ctx, size_t bytes, void **device_ptr);
int clr_buffer_free(clr_ctx *ctx, void *device_ptr);
int clr_buffer_upload(clr_ctx *ctx, void *device_ptr, const void *host, size_t bytes, void *stream);
int clr_buffer_download(clr_ctx *ctx, void *host, const void *device_ptr, size_t bytes, void *stream);
int clr_buffer_fill32(clr_ctx *ctx, void *device_ptr, uint32_t value, size_t count, void *stream);
/* replaces Accelerator.device_memcpy (accelerator.py:224-229): device-to-device copy of `bytes` */
int clr_buffer_copy(clr_ctx *ctx, void *dest, const void *src, size_t bytes, void *stream);

/* ---- replaces Accelerator.find_non_negative (accelerator.py:196-222, pyopencl GenericScanKernel):
 * stable compaction of 32-bit integers.  Writes the elements of in[0..n) that are >= 0 to
 * out[0..count) in their order and returns count in *count (host memory; the call waits for the
 * result like the reference's).  out must hold n elements in the worst case and must not overlap
 * in.  Three launches: per-tile counts (4096 elements a tile), scan of the counts, scatter; the
 * counts live in a work buffer the context keeps and grows on demand (4 B per tile). */
int clr_find_non_negative(clr_ctx *ctx, const int32_t *in, int32_t *out, size_t n, int *count, void *stream);

/* ---- generic launch: replaces Accelerator.call (accelerator.py:63-89) for step-wise kernels.
 * args[i] points to the i-th kernel argument.  If elapsed_ms is not NULL the launch is
 * bracketed by HIP events on `stream` and the call waits for it. */
int clr_launch(clr_program *prog, const char *kernel, const unsigned grid[3],
               const unsigned block[3], int n_args, void **args, void *stream, float *elapsed_ms);

/* ---- timing a region of launches on a stream with one HIP event pair: the per-kernel device
 * times of Accelerator.call / output_profiling_info (accelerator.py:83-89, 155-166) for launches
 * that go out back to back without a host wait in between.  clr_timer_end records the end and
 * returns at once; clr_timer_elapsed waits for it. */
int clr_timer_begin(clr_ctx *ctx, void *stream);
int clr_timer_end(clr_ctx *ctx, void *stream);
int clr_timer_elapsed(clr_ctx *ctx, float *elapsed_ms);

/* ---- the hot path: replaces the whole of Renderer.render_sample + compute_next_path_segment
 * (renderer.py:293-433) for `n_passes` consecutive sample passes. */
typedef struct clr_render_args {
    size_t struct_size;        /* sizeof(clr_render_args) of the caller's binding: a binding made from
                                * an older header fails cleanly instead of being read past its end */
    void *img;                 /* n_pixels x float4, running sum per pixel (blocked pixel order) */
    const void *cam;           /* n_pixels x float4, un-jittered primary ray per pixel */
    const void *pass_tape;     /* n_passes x 6 float4: camera sample per pass */
    const void *bounce_tape;   /* n_passes x max_bounces x 5 float4: random tape per segment
                                * (NULL for a program built for per-pixel random numbers) */
    const void *material_rows; /* spectral mode: n_passes x 10*(n_objects+1) floats; else NULL */
    const void *vector_data;   /* mesh vertices / normals (float4) or NULL */
    const void *integer_data;  /* mesh faces / octree stream (int32) or NULL */
    int n_pixels;
    int n_passes;
    int regenerate;            /* 1: a lane starts its next pass as soon as its path ends; must say which
                                * form of clr_render the program holds (module global clr_regen_mode) */
    void *debug_dump;          /* optional: max_bounces x n_pixels x 6 float4 (n_passes must be 1) */
    /* per-pixel random numbers (programs generated with rng_mode 'philox'; no counterpart in the
     * reference, whose random numbers are per frame): Philox key, global index of the first pass
     * of this call and global ray index of pixel 0, which together with the segment number form
     * the counter.  per_pixel_rng must say which kind of program `prog` is. */
    int per_pixel_rng;
    unsigned rng_seed_lo, rng_seed_hi;
    int first_pass, first_ray;
    /* pass-split programs (generated for small frames, where one lane per pixel leaves most of
     * the GPU idle; no counterpart in the reference): the passes of a pixel are dealt out to
     * `pass_groups` lanes, which log their addends to pass_log (n_passes x (1 + 6 max_bounces) x
     * n_pixels floats, device memory), and a second kernel adds the log to img in pass order -
     * the same additions in the same order as the one-lane walk.  pass_groups >= 1 and a
     * non-NULL pass_log are required for such a program and must be 0 / NULL for any other. */
    void *pass_log;
    int pass_groups;
    /* per-pixel random numbers when the passes are striped over several GPUs: local pass j of
     * this call has the global index first_pass + j * pass_stride (0 is read as 1), so that every
     * GPU draws exactly the numbers a single GPU would draw for the same pass */
    int pass_stride;
    /* pass-split programs, optional (NULL: as before): a stream whose work enqueued so far the
     * second kernel of this call waits for, so that consecutive calls overlap.  A pixel's sums
     * must be added in call order, but the first kernel of call k + 1 needs nothing of call k when
     * it writes to another pass_log.  A caller that deals its calls out to D (stream, pass_log)
     * pairs in turn and passes the previous call's stream here has up to D first kernels in
     * flight and the additions in call order (call k + D follows call k on the same stream, so
     * its log is free again); results do not change.  No counterpart in the reference. */
    void *reduce_after;
    /* per-pixel random numbers on a tile made of every N-th band of the frame's rays (several GPUs,
     * interleaved tile partition; no counterpart in the reference).  ray_band = 0: the tile is the
     * contiguous range of rays that starts at first_ray.  ray_band > 0: first_ray is the index
     * INSIDE the tile of pixel 0 of this call, and ray L of the tile is ray
     * ray_band_base + (L / ray_band) * ray_band_stride + L % ray_band of the frame. */
    int ray_band, ray_band_stride, ray_band_base;
} clr_render_args;

int clr_render_passes(clr_program *prog, const clr_render_args *args, void *stream,
                      float *elapsed_ms);

/* Device times of the two kernels of the last TIMED clr_render_passes call of this context on a
 * pass-split program (elapsed_ms covers both): clr_render, which logs the addends, and
 * clr_reduce_log, which adds them.  The per-kernel split of the reference's profiling table
 * (accelerator.py:83-89, 155-166) for the one call that launches two kernels.  Either may be NULL. */
int clr_last_split_ms(clr_ctx *ctx, float *render_ms, float *reduce_ms);

/* The same call for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * For pass-split programs only: the addend log - the one buffer of the path that is neither input
 * nor output - lives in a CLR_STATIC_LOG_BYTES array inside the library (one per GPU, held until
 * the process ends) instead of args->pass_log, which must be NULL.  The call allocates and frees
 * nothing; it fails when n_passes x (1 + 6 max_bounces) x n_pixels x 4 bytes exceed the array
 * (render fewer passes per call).  Results equal clr_render_passes bit for bit.  Every call on a
 * GPU uses that GPU's one array: a call is refused while the previous call on that GPU, made on
 * ANOTHER stream, is still running (calls on one stream are ordered by the stream); callers that
 * share the array between threads serialise their calls.  The array's address is looked up per
 * device in clr_ctx_create; if that fails there, this entry point reports why. */
#define CLR_STATIC_LOG_BYTES (128u << 20)
int clr_render_passes_static_log(clr_program *prog, const clr_render_args *args, void *stream,
                                 float *elapsed_ms);

/* First-hit AOV for the primary rays of the first pass record: ids = (object, sub-object) pairs. */
int clr_first_hit(clr_program *prog, const void *cam, const void *pass_tape,
                  const void *vector_data, const void *integer_data, int n_pixels,
                  void *out_ids, void *out_t, void *out_normal, void *stream);

/* ---- the exchange step of a multi-GPU render for a host that drives several GPUs from ONE
 * process (one clr_ctx per GPU), over RCCL / xGMI.  No counterpart in the reference, which asserts
 * a single device (accelerator.py:24-25); the path shards because the image is a sum over passes
 * (renderer.py:24-28).  A communicator spans the devices of `n` contexts (at most one per GPU).
 * clr_allreduce_sum_f32 sums `count` floats in place across buffers[i] (device memory of
 * ctxs[i]'s GPU): the per-GPU accumulators of a pass-striped render.  clr_allgather_f32
 * concatenates count_per_rank floats of send[i] into every recv[i] (n x count_per_rank floats):
 * the disjoint tiles of a tile-partitioned render.  streams: one hipStream_t per rank or NULL
 * for the default streams; the calls enqueue and return (clr_sync to wait).  The calling thread's
 * current HIP device is the same after a call as before it.  An error of one rank's collective
 * is reported after the group has been closed (the group's own error is then dropped).  RCCL is loaded on
 * the first call (dlopen); hosts with one process per GPU use torch.distributed instead
 * (raytracer_amd/multi_gpu.py). */
typedef struct clr_comm clr_comm;
int clr_comm_create(clr_ctx *const *ctxs, int n, clr_comm **out);
int clr_comm_destroy(clr_comm *comm);
int clr_comm_size(const clr_comm *comm, int *n);
int clr_allreduce_sum_f32(clr_comm *comm, void *const *buffers, size_t count, void *const *streams);
int clr_allgather_f32(clr_comm *comm, const void *const *send, void *const *recv, size_t count_per_rank,
                      void *const *streams);

/* ---- host-side octree builder: replaces Octree.build + get_data (objects/octree.py:69-156)
 * for meshes the reference's dense numpy construction cannot handle.  No device needed.
 * vertices: n_vertices x 3 floats, faces: n_faces x 3 ints. */
typedef struct clr_octree clr_octree;
typedef struct clr_octree_info_t {
    int n_nodes, n_leaves, max_depth;
    int stream_length;        /* ints in the reference-format node stream */
    int root_data_offset;     /* position of the root header in that stream */
    int flat_triangles;       /* (leaf, face) records in the flattened layout */
    double root_origin[3];
    double root_size;
} clr_octree_info_t;

int clr_octree_build(const float *vertices, int n_vertices, const int *faces, int n_faces,
                     int max_depth, int max_faces_per_leaf, clr_octree **out);
/* The same with a choice of the rule that files a face under a node.  overlap_rule 0: the
 * reference's (objects/octree.py:201-233: bounding sphere of the face against bounding sphere of
 * the node, and the node sphere must reach the face's plane), which reproduces the reference's
 * tree and stream.  overlap_rule 1: the triangle itself against the node's box (separating-axis
 * test): the same hits from a tree with a fraction of the leaf records, for large meshes. */
int clr_octree_build_rule(const float *vertices, int n_vertices, const int *faces, int n_faces,
                          int max_depth, int max_faces_per_leaf, int overlap_rule, clr_octree **out);
int clr_octree_info(const clr_octree *oct, clr_octree_info_t *info);
/* Copies out whichever destinations are not NULL: the reference stream (stream_length ints),
 * the flattened nodes (2 ints per node) and triangle records (12 floats per record). */
int clr_octree_copy(const clr_octree *oct, int *stream, int *flat_nodes, float *flat_tris);
int clr_octree_destroy(clr_octree *oct);

/* ---- mesh files (host only): replace read_off / read_zipper / remove_duplicate_faces
 * (mesh_formats.py:14-40, 42-70, 72-82).  format: 0 = OFF, 1 = Stanford "zipper" ASCII PLY.
 * One record per line; polygons are fan-triangulated around their first vertex; vertices are
 * the first three numbers of a vertex record, as doubles.  clr_mesh_unique_faces writes the
 * indices of the faces to keep (first occurrence of each SET of vertex indices). */
typedef struct clr_mesh clr_mesh;
int clr_mesh_read(const char *path, int format, clr_mesh **out);
int clr_mesh_info(const clr_mesh *mesh, int *n_vertices, int *n_triangles);
int clr_mesh_copy(const clr_mesh *mesh, double *vertices, int *triangles);
int clr_mesh_destroy(clr_mesh *mesh);
int clr_mesh_unique_faces(const int *faces, int n_faces, int *kept, int *n_kept);

#ifdef __cplusplus
}
#endif
#endif
