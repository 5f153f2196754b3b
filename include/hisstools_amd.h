/* hisstools_amd.h — C ABI of the MI355X-native partitioned-convolution engine.
 *
 * This is the drop-in boundary for the HISSTools_Library hot path
 *     Convolver -> NToMonoConvolve -> MonoConvolve -> {PartitionedConvolve, TimeDomainConvolve} -> HISSTools_FFT
 * The reference has no FFI of its own: hosts compile its C++ headers.  Every entry point below
 * therefore replaces exactly one public method of one reference class (cited as file:line relative to the
 * reference tree); the header-only C++ classes in include/hisstools_amd/ *.h re-create the reference's class
 * names and signatures on top of this ABI, so callers recompile unchanged.
 *
 * Conventions
 *   - plain pointers and sizes only; handles are opaque; no C++ or torch types cross the boundary
 *   - int return values named "error" are ConvolveError codes (ConvolveErrors.h:4-19), reproduced below
 *   - all audio / IR data is IEEE float32 (double overloads convert, as Convolver.cpp:126-134,156-183)
 *   - *_dev variants take pointers to memory on the engine's GPU and never touch host memory (HBM-resident use)
 *   - a NULL handle from a *_create means no usable GPU / out of memory; hcv_last_error() says why.  There is no
 *     CPU fallback.
 */
#ifndef HISSTOOLS_AMD_H
#define HISSTOOLS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HCV_API __attribute__((visibility("default")))

/* ConvolveErrors.h:4-19 */
enum
{
    HCV_ERR_NONE = 0,
    HCV_ERR_IN_CHAN_OUT_OF_RANGE = 1,
    HCV_ERR_OUT_CHAN_OUT_OF_RANGE = 2,
    HCV_ERR_MEM_UNAVAILABLE = 3,
    HCV_ERR_MEM_ALLOC_TOO_SMALL = 4,
    HCV_ERR_TIME_IMPULSE_TOO_LONG = 5,
    HCV_ERR_TIME_LENGTH_OUT_OF_RANGE = 6,
    HCV_ERR_PARTITION_LENGTH_TOO_LARGE = 7,
    HCV_ERR_FFT_SIZE_MAX_TOO_SMALL = 8,
    HCV_ERR_FFT_SIZE_MAX_TOO_LARGE = 9,
    HCV_ERR_FFT_SIZE_MAX_NON_POWER_OF_TWO = 10,
    HCV_ERR_FFT_SIZE_OUT_OF_RANGE = 11,
    HCV_ERR_FFT_SIZE_NON_POWER_OF_TWO = 12
};

/* LatencyMode, MonoConvolve.h:14-19 */
enum { HCV_LATENCY_ZERO = 0, HCV_LATENCY_SHORT = 1, HCV_LATENCY_MEDIUM = 2 };

typedef struct hcv_partitioned hcv_partitioned;   /* HISSTools::PartitionedConvolve */
typedef struct hcv_timedomain hcv_timedomain;     /* HISSTools::TimeDomainConvolve  */
typedef struct hcv_mono hcv_mono;                 /* HISSTools::MonoConvolve        */
typedef struct hcv_ntomono hcv_ntomono;           /* HISSTools::NToMonoConvolve     */
typedef struct hcv_convolver hcv_convolver;       /* HISSTools::Convolver           */

/* ---------------------------------------------------------------- library / device */

HCV_API const char *hcv_version(void);
HCV_API int hcv_device_count(void);                       /* number of HIP devices, 0 if none / no runtime */
HCV_API int hcv_set_default_device(int device);           /* device used by subsequently created objects */
HCV_API int hcv_get_default_device(void);
HCV_API const char *hcv_last_error(void);                 /* thread-local text of the last failure */
/* The control arena (MI355X extension; the reference's MemorySwap allocates on the control thread, MemorySwap.h:187-229, and its malloc
 * never touches the audio thread): device memory for set / resize — regrown spectra, staging buffers — is carved out of an arena per
 * device that is mapped when objects are CREATED, so that no later control call has the driver map memory under a running audio thread
 * (which stalls every HIP call of the process for tens of milliseconds).  Every object asks for what one regrow of its largest stage by
 * half needs (16 MiB .. 1 GiB); the arena shrinks again as objects go and is released with the device's last one.  A host that will load
 * impulse responses far longer than its objects were created for reserves more: hcv_ctl_reserve(device, bytes) — any time; an existing
 * arena grows at once — or HCV_CTL_RESERVE_MB set the least the device holds while it has an object.  hcv_ctl_reserved reports what it
 * holds now. */
HCV_API int hcv_ctl_reserve(int device, size_t bytes);
HCV_API size_t hcv_ctl_reserved(int device);
/* HCV_ORDER_CHECK=1 (debug aid, MI355X extension): while the engines enqueue, the order DESIGN.md section 2 states — program order on a
 * stream, event records and waits, ring depths — is followed with vector clocks and every declared buffer access checked against it;
 * a violation is reported on stderr.  Returns the number of violations so far in this process, -1 when the check is off. */
HCV_API long long hcv_order_check_violations(void);
/* Debug aid: on SIGSEGV / SIGABRT / SIGBUS print the faulting thread's native call stack to stderr (glibc backtrace), then hand the signal
 * on to whoever had it before (Python's faulthandler, the default action).  tests/conftest.py installs it when HCV_NATIVE_BACKTRACE=1. */
HCV_API void hcv_debug_native_backtrace_on_crash(void);

/* ---------------------------------------------------------------- HISSTools_FFT (float real transforms on the path)
 * hisstools_rfft 5-arg  HISSTools_FFT.cpp:226-230   (zero-padding unzip + real FFT, output x2, vDSP packing)
 * hisstools_rifft 4-arg HISSTools_FFT.cpp:244-248   (unnormalised; the input spectrum is NOT modified here)
 * Batched: `batch` rows, input row stride in_stride floats, spectra rows of 2^(log2n-1) floats each.
 * Returns 0 on success, -1 on failure (hcv_last_error). */
HCV_API int hcv_rfft_f32(const float *in, size_t in_length, size_t in_stride, size_t batch, unsigned log2n, float *realp, float *imagp);
HCV_API int hcv_rifft_f32(const float *realp, const float *imagp, size_t batch, unsigned log2n, float *out);

/* ---------------------------------------------------------------- PartitionedConvolve (PartitionedConvolve.h:23-41) */

HCV_API hcv_partitioned *hcv_partitioned_create(uintptr_t maxFFTSize, uintptr_t maxLength, uintptr_t offset, uintptr_t length); /* .cpp:52-102 */
HCV_API void hcv_partitioned_destroy(hcv_partitioned *h);                                   /* .cpp:104-112 */
HCV_API int hcv_partitioned_set_fft_size(hcv_partitioned *h, uintptr_t FFTSize);             /* .cpp:131-154 error */
HCV_API int hcv_partitioned_set_length(hcv_partitioned *h, uintptr_t length);                /* .cpp:156-161 error */
HCV_API void hcv_partitioned_set_offset(hcv_partitioned *h, uintptr_t offset);               /* .cpp:163-166 */
HCV_API void hcv_partitioned_set_reset_offset(hcv_partitioned *h, intptr_t offset);          /* .cpp:168-171 (phase hint) */
HCV_API int hcv_partitioned_set(hcv_partitioned *h, const float *input, uintptr_t length);   /* .cpp:173-225 error */
HCV_API void hcv_partitioned_reset(hcv_partitioned *h);                                      /* .cpp:227-230 */
HCV_API int hcv_partitioned_process(hcv_partitioned *h, const float *in, float *out, uintptr_t numSamples); /* .cpp:243-385; 1 = out written, 0 = untouched, -1 = device failure */

/* ---------------------------------------------------------------- TimeDomainConvolve (TimeDomainConvolve.h:15-31) */

HCV_API hcv_timedomain *hcv_timedomain_create(uintptr_t offset, uintptr_t length);           /* .cpp:33-50 */
HCV_API void hcv_timedomain_destroy(hcv_timedomain *h);
HCV_API int hcv_timedomain_set_length(hcv_timedomain *h, uintptr_t length);                  /* .cpp:62-67 error */
HCV_API void hcv_timedomain_set_offset(hcv_timedomain *h, uintptr_t offset);                 /* .cpp:57-60 */
HCV_API int hcv_timedomain_set(hcv_timedomain *h, const float *input, uintptr_t length);     /* .cpp:69-87 error */
HCV_API void hcv_timedomain_reset(hcv_timedomain *h);                                        /* .cpp:89-92 */
HCV_API int hcv_timedomain_process(hcv_timedomain *h, const float *in, float *out, uintptr_t numSamples); /* .cpp:128-163; returns "has taps" */

/* ---------------------------------------------------------------- MonoConvolve (MonoConvolve.h:30-48) */

HCV_API hcv_mono *hcv_mono_create(uintptr_t maxLength, int latency);                         /* .cpp:18-32 */
/* custom partitioning, .cpp:36-45; where the reference throws std::runtime_error this returns NULL and copies the
 * message ("invalid FFT size or order" / "no valid FFT sizes given") into err */
HCV_API hcv_mono *hcv_mono_create_custom(uintptr_t maxLength, int zeroLatency, uint32_t A, uint32_t B, uint32_t C, uint32_t D, char *err, size_t errlen);
HCV_API void hcv_mono_destroy(hcv_mono *h);
HCV_API void hcv_mono_set_reset_offset(hcv_mono *h, intptr_t offset);                        /* .cpp:80-99 */
HCV_API int hcv_mono_resize(hcv_mono *h, uintptr_t length);                                  /* .cpp:101-110 error */
HCV_API int hcv_mono_set(hcv_mono *h, const float *input, uintptr_t length, int requestResize); /* .cpp:118-140 error */
HCV_API int hcv_mono_reset(hcv_mono *h);                                                     /* .cpp:148-152 error */
HCV_API int hcv_mono_process(hcv_mono *h, const float *in, float *temp, float *out, uintptr_t numSamples, int accumulate); /* .cpp:179-201; 1 = out written/added, 0 = untouched, -1 failure */

/* ---------------------------------------------------------------- NToMonoConvolve (NToMonoConvolve.h:18-24) */

HCV_API hcv_ntomono *hcv_ntomono_create(uint32_t inChans, uintptr_t maxLength, int latency); /* .cpp:4-9 */
HCV_API void hcv_ntomono_destroy(hcv_ntomono *h);
HCV_API int hcv_ntomono_resize(hcv_ntomono *h, uint32_t inChan, uintptr_t length);           /* .cpp:20-23 error */
HCV_API int hcv_ntomono_set(hcv_ntomono *h, uint32_t inChan, const float *input, uintptr_t length, int resize); /* .cpp:25-28 error */
HCV_API int hcv_ntomono_reset(hcv_ntomono *h, uint32_t inChan);                              /* .cpp:30-33 error */
HCV_API int hcv_ntomono_process(hcv_ntomono *h, const float *const *ins, float *out, float *temp, size_t numSamples, size_t activeInChans); /* .cpp:35-43; 0 ok, -1 failure */

/* ---------------------------------------------------------------- Convolver (Convolver.h:23-50) */

HCV_API hcv_convolver *hcv_convolver_create(uint32_t numIns, uint32_t numOuts, int latency); /* Convolver.cpp:5-22 */
HCV_API hcv_convolver *hcv_convolver_create_parallel(uint32_t numIO, int latency);           /* Convolver.cpp:24-41 */
HCV_API void hcv_convolver_destroy(hcv_convolver *h);                                        /* Convolver.cpp:43-47 */
HCV_API void hcv_convolver_clear(hcv_convolver *h, int resize);                              /* :51-64 */
HCV_API void hcv_convolver_clear_chan(hcv_convolver *h, uint32_t inChan, uint32_t outChan, int resize); /* :66-69 */
HCV_API void hcv_convolver_reset(hcv_convolver *h);                                          /* :73-86 */
HCV_API int hcv_convolver_reset_chan(hcv_convolver *h, uint32_t inChan, uint32_t outChan);   /* :88-98 error */
HCV_API int hcv_convolver_resize(hcv_convolver *h, uint32_t inChan, uint32_t outChan, uintptr_t length); /* :100-110 error */
HCV_API int hcv_convolver_set_f32(hcv_convolver *h, uint32_t inChan, uint32_t outChan, const float *input, uintptr_t length, int resize);  /* :114-124 error */
HCV_API int hcv_convolver_set_f64(hcv_convolver *h, uint32_t inChan, uint32_t outChan, const double *input, uintptr_t length, int resize); /* :126-134 error */
HCV_API int hcv_convolver_process_f32(hcv_convolver *h, const float *const *ins, float **outs, size_t numIns, size_t numOuts, size_t numSamples);   /* :138-154; 0 ok, -1 failure */
HCV_API int hcv_convolver_process_f64(hcv_convolver *h, const double *const *ins, double **outs, size_t numIns, size_t numOuts, size_t numSamples); /* :156-183 */

/* ---------------------------------------------------------------- MI355X extensions of the Convolver (no reference analogue)
 * Multi-GPU: one object per GPU, each owning a block of output rows (no collective on the data path); see
 * hisstools_library_amd/sharded.py.  `device` < 0 = default device. */
HCV_API hcv_convolver *hcv_convolver_create_on(uint32_t numIns, uint32_t numOuts, int latency, int device, uint32_t maxBlock);
HCV_API hcv_convolver *hcv_convolver_create_custom(uint32_t numIns, uint32_t numOuts, int parallel, uintptr_t maxLength, int zeroLatency,
                                                   uint32_t A, uint32_t B, uint32_t C, uint32_t D, int device, uint32_t maxBlock);
/* As create_custom, plus an extended non-uniform partitioning of the far tail: past the reference's largest FFT the IR
 * is served by FFTs `tailRatio` (2, 4 or 8; 0 = off) times larger per rung, up to 2^20, as far as maxLength reaches.
 * Same convolution, same latency; the far tail moves tailRatio x fewer HBM bytes per sample per rung. */
HCV_API hcv_convolver *hcv_convolver_create_extended(uint32_t numIns, uint32_t numOuts, int parallel, uintptr_t maxLength, int zeroLatency,
                                                     uint32_t A, uint32_t B, uint32_t C, uint32_t D, int device, uint32_t maxBlock,
                                                     uint32_t tailRatio);
/* IR already in HBM on the object's device */
HCV_API int hcv_convolver_set_f32_dev(hcv_convolver *h, uint32_t inChan, uint32_t outChan, const float *input_dev, uintptr_t length, int resize);
/* ins_dev: [numIns][in_stride] floats, outs_dev: [numOuts][out_stride] floats, both in HBM.  Asynchronous on the object's
 * streams unless sync != 0: the input rows must hold their samples when the call is made (what produced them has finished, or was
 * synchronised with), and both buffers are the object's until hcv_convolver_synchronize — or a call with sync != 0 — returns: the
 * block's launches read and write them on streams of the object's own (one block's transforms may run beside the previous block's
 * inverse).  Returns 0 ok, -1 failure. */
HCV_API int hcv_convolver_process_f32_dev(hcv_convolver *h, const float *ins_dev, size_t in_stride, float *outs_dev, size_t out_stride,
                                          size_t numIns, size_t numOuts, size_t numSamples, int sync);
HCV_API int hcv_convolver_synchronize(hcv_convolver *h);
HCV_API int hcv_convolver_device(hcv_convolver *h);

/* Multi-GPU, one host object (Convolver.h:23-50 stays the interface): `numDevices` engines, one per listed device
 * (an index may repeat: several shards on one GPU).  Output rows are split over the devices first — no exchange, each device
 * delivers final samples for its rows.  With fewer output rows than devices the inputs are split as well and the partial blocks
 * of a row group are summed on the group's first device: the per-output sum of NToMonoConvolve.cpp:39-42 taken across GPUs
 * (peer reads over xGMI).  Every hcv_convolver_* call works on the result: set / resize / clear / reset go to the shard that
 * owns the pair, process splits the channel pointers (host path: all shards are begun before the first is waited for; device
 * path: buffers on devices[0], read and written in place by every device).  Parameters as hcv_convolver_create_custom.
 * HCV_DEVICES="0,1,..." in the environment makes hcv_convolver_create / _create_parallel (and so HISSTools::Convolver) build
 * such an object, so that a caller that recompiles unchanged uses every GPU listed. */
HCV_API hcv_convolver *hcv_convolver_create_sharded(uint32_t numIns, uint32_t numOuts, int parallel, uintptr_t maxLength, int zeroLatency,
                                                    uint32_t A, uint32_t B, uint32_t C, uint32_t D, const int *devices, int numDevices,
                                                    uint32_t maxBlock);
HCV_API int hcv_convolver_num_shards(hcv_convolver *h);

/* One process per GPU (torch.distributed, MPI): each rank owns a single-device object holding its share of the inputs of a row
 * group, and the group's partial output blocks are summed with ONE ncclAllReduce(ncclFloat, ncclSum) per call, enqueued on the
 * object's own stream behind the block (RCCL over xGMI; librccl is bound at run time, the copy the process already holds is
 * used).  hcv_rccl_unique_id fills 128 bytes (ncclUniqueId) on one rank; the caller distributes them (its launcher's store);
 * every rank of the group then calls hcv_convolver_comm_init with its rank in the group.  _allreduce = process_f32_dev + the
 * in-place all-reduce of outs_dev (numOuts rows of numSamples).  0 ok, -1 failure. */
HCV_API int hcv_rccl_unique_id(void *out128);
HCV_API int hcv_convolver_comm_init(hcv_convolver *h, const void *unique_id128, int rank, int nranks);
HCV_API int hcv_convolver_process_f32_dev_allreduce(hcv_convolver *h, const float *ins_dev, size_t in_stride, float *outs_dev, size_t out_stride,
                                                    size_t numIns, size_t numOuts, size_t numSamples, int sync);

/* Registered host memory (optional).  The reference's process() takes host channel pointers (Convolver.cpp:138-154); by default
 * every call here stages them through pinned buffers (two host copies of the block, PCIe both ways).  A caller that keeps its
 * channel buffers for a while — a plug-in host — can pin and map them ONCE: hcv_convolver_process_f32 calls whose input rows and
 * output rows each form one evenly spaced block inside registered memory then run on the caller's memory in place (the kernels
 * read and write it over PCIe), with no staging copy.  Anything else takes the staging path as before.  Unregister before the
 * memory is freed.  0 ok, -1 failure (hcv_last_error). */
HCV_API int hcv_host_register(void *ptr, size_t bytes);
HCV_API int hcv_host_unregister(void *ptr);

/* The audio-thread contract (MemorySwap::attempt, MemorySwap.h:182-185; MonoConvolve.cpp:181-183; ThreadLocks.hpp:51-87): process never
 * waits for a control call — there is no lock on its path.  The engine's host state has an OWNER: the thread inside a process call, or,
 * only while no stream is running (no process call for 0.4 s), a control thread inside the short section that swaps its staged result
 * in.  While a stream is running control calls post that section and the audio thread runs it between two of its blocks; the pair
 * plays its previous impulse response until then (the reference mutes it).  Counters since the last hcv_convolver_clear_stats. */
typedef struct hcv_rt_stats
{
    uint64_t start_collisions;  /* process calls that found a control thread inside a section: only the FIRST call of a stream (after a pause of
                                 * 0.4 s or more) can; that call's block is silent — no wait, no retry — and the next call proceeds (calls of 2048 samples or more: start_waits) */
    uint64_t mailbox_runs;      /* swap sections of control calls that the audio thread ran between two of its blocks */
    uint64_t mailbox_ns_max;    /* ... the longest of them, nanoseconds of the audio thread's time */
    uint64_t mailbox_ns_total;  /* ... and their sum */
    uint64_t ctl_sections;      /* swap sections control threads ran themselves (no stream running) */
    uint64_t start_waits;       /* calls of 2048 samples or more (an offline loop, not an audio callback) that met such a section and waited it out —
                                 * sleeping, 100 ms at most — instead of delivering a silent block of that size */
    uint64_t arena_misses;      /* control-path allocations the control arena could not serve: the driver mapped memory, and every stream of the
                                 * process may have stalled for tens of milliseconds meanwhile (hcv_ctl_reserve) */
} hcv_rt_stats;
HCV_API int hcv_convolver_rt_stats(hcv_convolver *h, hcv_rt_stats *out);

/* per-stage measurements of the spectral multiply-accumulate kernel (HIP events on the launch stream) */
typedef struct hcv_stage_stats
{
    uint32_t fft_size, partitions, num_ins, num_outs;
    uint64_t mac_launches, mac_hops;
    double mac_ms;
    uint32_t ksplit, out_tile;
    /* launches of the steady-state instantiation (every partition of every pair live: no per-pair bounds, IR spectra
     * streamed with nontemporal loads) among mac_launches; the hop tile of the last launch; the partitions it reduced over
     * (whole-hop blocks: the stage's own + the one holding the IR in front of its segment) */
    uint64_t mac_steady_launches;
    uint32_t hop_tile, launch_partitions;
    /* of mac_launches: whole blocks of a one-output engine that ran as ONE launch (transforms, multiply-accumulate and inverse
     * with in-launch hand-overs) */
    uint64_t fused_launches;
    /* times the stage stood its n x m block down (64 blocks, longer — up to 4096 — when it recurs): three launches within 64 blocks found their forward launch missing
     * (stuck behind another stream's work in a shared hardware queue) and did the transforms themselves */
    uint64_t fused_stood_down;
    uint64_t host_pre_launches;     /* hop-sized host-pointer calls whose partitions >= 1 were multiplied ahead of the upload (streamed engines) */
} hcv_stage_stats;
HCV_API void hcv_convolver_set_profiling(hcv_convolver *h, int on);
HCV_API int hcv_convolver_num_stages(hcv_convolver *h);
HCV_API int hcv_convolver_stage_stats(hcv_convolver *h, int stage, hcv_stage_stats *out);
HCV_API void hcv_convolver_clear_stats(hcv_convolver *h);

/* ---------------------------------------------------------------- spectral_processor<float> (first "next" row)
 * One-shot FFT convolution / correlation of two real signals with edge modes — the real overloads of
 * spectral_processor::convolve / correlate (SpectralProcessor.hpp:173-184; sizes :210-218,549-560).
 * mode: 0 Linear, 1 Wrap, 2 WrapCentre, 3 Fold, 4 FoldRepeat (EdgeMode, SpectralProcessor.hpp:22).
 * hcv_spectral_size = convolved_size = correlated_size: samples written to `out` (0 = nothing is written: an empty input,
 * or an FFT beyond 2^22 — the engine's "max_fft_size"; the _dev entries below stop at 2^20).  Returns 0 ok, -1 device failure. */
HCV_API size_t hcv_spectral_size(size_t size1, size_t size2, int mode);
HCV_API int hcv_spectral_convolve_f32(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out);
HCV_API int hcv_spectral_correlate_f32(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out);
/* the same with all three buffers resident in HBM (device pointers), enqueued on `stream` (a hipStream_t, NULL = default);
 * `out` must hold hcv_spectral_size(...) floats.  Scratch is cached per device; overlapping calls on different streams must
 * be ordered by the caller. */
HCV_API int hcv_spectral_convolve_f32_dev(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out, void *stream, int sync);
HCV_API int hcv_spectral_correlate_f32_dev(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out, void *stream, int sync);

/* the real overloads in double and the complex overloads in float and double (SpectralProcessor.hpp:164-167, 176-179; binary_op
 * :559-614).  A complex operand is a real and an imaginary array that may differ in length (the shorter is zero-padded); its
 * size is the longer of the two, and r_out / i_out each receive hcv_spectral_size(size1, size2, mode) values.  In the two wrap
 * modes the complex overloads follow the real overloads' arrangement: the reference's complex instantiation reads past the
 * result there (its Split wrap() takes an offset where the shared arrange code passes an end position, :401-408 / :429-435). */
HCV_API int hcv_spectral_convolve_f64(const double *in1, size_t size1, const double *in2, size_t size2, int mode, double *out);
HCV_API int hcv_spectral_correlate_f64(const double *in1, size_t size1, const double *in2, size_t size2, int mode, double *out);
HCV_API int hcv_spectral_convolve_complex_f32(const float *r_in1, size_t r_size1, const float *i_in1, size_t i_size1, const float *r_in2, size_t r_size2,
                                              const float *i_in2, size_t i_size2, int mode, float *r_out, float *i_out);
HCV_API int hcv_spectral_correlate_complex_f32(const float *r_in1, size_t r_size1, const float *i_in1, size_t i_size1, const float *r_in2, size_t r_size2,
                                               const float *i_in2, size_t i_size2, int mode, float *r_out, float *i_out);
HCV_API int hcv_spectral_convolve_complex_f64(const double *r_in1, size_t r_size1, const double *i_in1, size_t i_size1, const double *r_in2, size_t r_size2,
                                              const double *i_in2, size_t i_size2, int mode, double *r_out, double *i_out);
HCV_API int hcv_spectral_correlate_complex_f64(const double *r_in1, size_t r_size1, const double *i_in1, size_t i_size1, const double *r_in2, size_t r_size2,
                                               const double *i_in2, size_t i_size2, int mode, double *r_out, double *i_out);

/* ---------------------------------------------------------------- the full hisstools_* FFT surface (second "next" row)
 * Every transform of HISSTools_FFT.h:87-369 — float and double, complex and real, in place on split data or out of
 * place from / to interleaved samples, plus zip / unzip — as ONE batched entry point.  `op` selects the reference
 * function, `precision` its overload:
 *
 *   HCV_FFT_FFT        hisstools_fft    HISSTools_FFT.h:130,142   complex forward, split (a = realp, b = imagp), 2^log2n points
 *   HCV_FFT_IFFT       hisstools_ifft   HISSTools_FFT.h:220,232   complex inverse (unnormalised)
 *   HCV_FFT_RFFT       hisstools_rfft   HISSTools_FFT.h:154,166   real forward on unzipped data: 2^(log2n-1) split values in,
 *                                                                 packed half spectrum out (x2, DC/Nyquist in bin 0)
 *   HCV_FFT_RIFFT      hisstools_rifft  HISSTools_FFT.h:244,256   real inverse, split in / split (unzipped samples) out, unnormalised
 *   HCV_FFT_RFFT_ZIP   hisstools_rfft   HISSTools_FFT.h:180,194,208  in_length samples (src_a), zero padded to 2^log2n -> packed spectrum
 *   HCV_FFT_RIFFT_ZIP  hisstools_rifft  HISSTools_FFT.h:269,282   packed spectrum -> 2^log2n samples (dst_a)
 *   HCV_FFT_UNZIP      hisstools_unzip / hisstools_unzip_zero  HISSTools_FFT.h:295-345  (in_length = 2^log2n for plain unzip)
 *   HCV_FFT_ZIP        hisstools_zip    HISSTools_FFT.h:357,369
 *
 * Sizes: complex log2n 0..22, real log2n 0..23 (the reference's FFT_Tester sweeps 0..21), zip/unzip up to 2^30.  `batch`
 * transforms are processed per call; consecutive transforms lie src_stride / dst_stride ELEMENTS apart (0 = densely
 * packed).  Operands may alias exactly (in place) or not at all.  The forward and inverse transforms are the
 * reference's: unnormalised, real spectra doubled, so rifft(rfft(x)) = 2N x and ifft(fft(z)) = N z.
 * Unlike the reference the out-of-place rifft leaves its input spectrum untouched.
 *
 * hcv_fft_exec takes host pointers (one upload, the kernels, one download); hcv_fft_exec_dev takes device pointers on
 * the default device and enqueues on `stream` (a hipStream_t, NULL = the default stream), waiting only if sync != 0.
 * Returns 0 ok, -1 on a bad descriptor or a device failure (hcv_last_error). */
enum { HCV_FFT_FFT = 0, HCV_FFT_IFFT = 1, HCV_FFT_RFFT = 2, HCV_FFT_RIFFT = 3, HCV_FFT_RFFT_ZIP = 4, HCV_FFT_RIFFT_ZIP = 5,
       HCV_FFT_UNZIP = 6, HCV_FFT_ZIP = 7 };
enum { HCV_FFT_F32 = 0, HCV_FFT_F64 = 1, HCV_FFT_F32_TO_F64 = 2 /* float samples in, double split out: RFFT_ZIP and UNZIP only */ };
typedef struct hcv_fft_call
{
    int op, precision;
    unsigned log2n;
    size_t batch;
    const void *src_a, *src_b;          /* split: realp / imagp; samples: src_a only */
    void *dst_a, *dst_b;
    size_t src_stride, dst_stride;
    size_t in_length;                   /* RFFT_ZIP / UNZIP: valid samples per transform */
} hcv_fft_call;
HCV_API int hcv_fft_exec(const hcv_fft_call *call);
HCV_API int hcv_fft_exec_dev(const hcv_fft_call *call, void *stream, int sync);

/* ---------------------------------------------------------------- spectral IR functions (fourth "next" row)
 * IR pre-processing in the spectral domain on batches of packed half spectra (fft_size = 2^log2n real samples,
 * fft_size / 2 values per array, bin 0 = (DC, Nyquist)) — SpectralFunctions.hpp:365-413:
 *
 *   HCV_IR_COPY          ir_copy          :365-369
 *   HCV_IR_SPIKE         ir_spike         :371-375   value = spike position in samples (src unused)
 *   HCV_IR_DELAY         ir_delay         :377-384   value = delay in samples (fractional allowed; circular)
 *   HCV_IR_TIME_REVERSE  ir_time_reverse  :386-390
 *   HCV_IR_PHASE         ir_phase         :392-413   value = phase, 0 minimum .. 0.5 linear .. 1 maximum; zero_center as the reference
 *
 * precision: HCV_FFT_F32 or HCV_FFT_F64.  src may equal dst.  Strides in elements, 0 = dense.  hcv_ir_exec takes host
 * pointers, hcv_ir_exec_dev device pointers + a hipStream_t.  Returns 0 ok, -1 failure (hcv_last_error).
 *
 * hcv_spectral_change_phase_* = spectral_processor<T>::change_phase (SpectralProcessor.hpp:188-208): real FFT of `size`
 * samples at the size hcv_spectral_phase_size(size, time_multiplier) returns, ir_phase, inverse FFT, scale 0.5 / fft_size;
 * `out` receives that many samples (1 for a single-sample input, which is copied). */
enum { HCV_IR_COPY = 0, HCV_IR_SPIKE = 1, HCV_IR_DELAY = 2, HCV_IR_TIME_REVERSE = 3, HCV_IR_PHASE = 4 };
typedef struct hcv_ir_call
{
    int op, precision;
    unsigned log2n;
    size_t batch;
    const void *src_re, *src_im;
    void *dst_re, *dst_im;
    size_t src_stride, dst_stride;
    double value;
    int zero_center;
} hcv_ir_call;
HCV_API int hcv_ir_exec(const hcv_ir_call *call);
HCV_API int hcv_ir_exec_dev(const hcv_ir_call *call, void *stream, int sync);
/* The IR products, SpectralFunctions.hpp:415-436:
 *
 *   HCV_IR_CONVOLVE_COMPLEX   ir_convolve_complex   :414-418   dst = scale * a * b        on `size` values per array
 *   HCV_IR_CONVOLVE_REAL      ir_convolve_real      :420-424   the same on packed half spectra of `size` real samples: size / 2 values
 *   HCV_IR_CORRELATE_COMPLEX  ir_correlate_complex  :426-430   dst = scale * a * conj(b)    per array, bin 0 = (DC, Nyquist) gets its two
 *   HCV_IR_CORRELATE_REAL     ir_correlate_real     :432-436                                real products
 *
 * `size` = the reference's fft_size argument, a power of two (its vector loops drop the remainder of any other size).  Every product and
 * sum is rounded on its own, then the scale, as the reference's SIMD layer does: bit-identical results.  b_broadcast != 0: ONE b
 * spectrum for the whole batch (a filter applied to many spectra).  dst may equal a or b.  Strides in elements, 0 = dense. */
enum { HCV_IR_CONVOLVE_COMPLEX = 0, HCV_IR_CONVOLVE_REAL = 1, HCV_IR_CORRELATE_COMPLEX = 2, HCV_IR_CORRELATE_REAL = 3 };
typedef struct hcv_ir_product_call
{
    int op, precision;
    size_t size, batch;
    const void *a_re, *a_im, *b_re, *b_im;
    void *dst_re, *dst_im;
    size_t a_stride, b_stride, dst_stride;
    int b_broadcast;
    double scale;
} hcv_ir_product_call;
HCV_API int hcv_ir_product_exec(const hcv_ir_product_call *call);                            /* host pointers */
HCV_API int hcv_ir_product_exec_dev(const hcv_ir_product_call *call, void *stream, int sync);   /* device pointers + a hipStream_t */
HCV_API size_t hcv_spectral_phase_size(size_t size, double time_multiplier);
HCV_API int hcv_spectral_change_phase_f32(const float *in, size_t size, double phase, double time_multiplier, float *out);
HCV_API int hcv_spectral_change_phase_f64(const double *in, size_t size, double phase, double time_multiplier, double *out);

/* ---------------------------------------------------------------- audio files (third "next" row; host-side only)
 * HISSTools::IAudioFile / OAudioFile (AudioFile/IAudioFile.h:37-54, OAudioFile.h:14-37, BaseAudioFile.h:18-98): WAVE
 * (RIFF / RIFX), AIFF and AIFC reading; WAVE and AIFC writing (an AIFF request writes AIFC, as the reference); PCM int
 * 8 / 16 / 24 / 32 and float 32 / 64.  Enumerations keep the reference's values:
 *   file type   0 none, 1 AIFF, 2 AIFC, 3 WAVE                      (BaseAudioFile.h:20-26)
 *   pcm format  0 int8, 1 int16, 2 int24, 3 int32, 4 float32, 5 float64   (:27-35)
 *   endianness  0 little, 1 big                                     (:36-40)
 *   error flags BaseAudioFile::Error bits                           (:47-66)
 * An open that fails still returns an object (never NULL): query hcv_audiofile_is_open / error_flags, as with the
 * reference's constructors.  Sample conversion and quantisation are the reference's (integers scaled by 2^(bits-1),
 * no clipping on write except WAVE 8-bit).  Differences: reads past the end deliver zeros, AIFC "fl64" is read as 64-bit
 * floats. */
typedef struct hcv_audiofile hcv_audiofile;
typedef struct hcv_audiofile_info
{
    int file_type, pcm_format, header_endianness, audio_endianness;
    double sampling_rate;
    unsigned channels, frames, bit_depth;
    int error_flags;
} hcv_audiofile_info;
HCV_API hcv_audiofile *hcv_iaudiofile_open(const char *path);                                       /* IAudioFile.cpp:23-50 */
HCV_API hcv_audiofile *hcv_oaudiofile_open(const char *path, int type, int format, unsigned channels, double sampling_rate,
                                           int endianness /* -1 = the type's default */);           /* OAudioFile.cpp:43-77 */
HCV_API void hcv_audiofile_close(hcv_audiofile *h);
HCV_API int hcv_audiofile_is_open(const hcv_audiofile *h);
HCV_API int hcv_audiofile_get_info(const hcv_audiofile *h, hcv_audiofile_info *out);
HCV_API void hcv_audiofile_seek(hcv_audiofile *h, uint32_t frame);                                   /* IAudioFile.cpp:70-73, OAudioFile.cpp:92-96 */
HCV_API uint32_t hcv_audiofile_position(hcv_audiofile *h);
HCV_API void hcv_iaudiofile_read_raw(hcv_audiofile *h, void *out, uint32_t frames);                  /* IAudioFile.cpp:85-88 */
HCV_API void hcv_iaudiofile_read_interleaved_f32(hcv_audiofile *h, float *out, uint32_t frames);     /* :95-98 */
HCV_API void hcv_iaudiofile_read_interleaved_f64(hcv_audiofile *h, double *out, uint32_t frames);    /* :90-93 */
HCV_API void hcv_iaudiofile_read_channel_f32(hcv_audiofile *h, float *out, uint32_t frames, unsigned channel);   /* :105-108 */
HCV_API void hcv_iaudiofile_read_channel_f64(hcv_audiofile *h, double *out, uint32_t frames, unsigned channel);  /* :100-103 */
HCV_API void hcv_oaudiofile_write_raw(hcv_audiofile *h, const void *in, uint32_t frames);            /* OAudioFile.h:29 */
HCV_API void hcv_oaudiofile_write_interleaved_f32(hcv_audiofile *h, const float *in, uint32_t frames);   /* OAudioFile.cpp:106-114 */
HCV_API void hcv_oaudiofile_write_interleaved_f64(hcv_audiofile *h, const double *in, uint32_t frames);
HCV_API void hcv_oaudiofile_write_channel_f32(hcv_audiofile *h, const float *in, uint32_t frames, unsigned channel);   /* :116-124 */
HCV_API void hcv_oaudiofile_write_channel_f64(hcv_audiofile *h, const double *in, uint32_t frames, unsigned channel);

#ifdef __cplusplus
}
#endif

#endif /* HISSTOOLS_AMD_H */
