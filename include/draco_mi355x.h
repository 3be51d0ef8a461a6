/* include/draco_mi355x.h
 *
 * C-ABI of libdraco_mi355x.so: batched Draco (bitstream 2.2) mesh decode on one
 * MI355X per context.  This is the drop-in boundary under draco-sharp's
 *     DracoDecoder.Decode(BinaryReader)            src/Draco/IO/DracoDecoder.cs:19-42
 * The reference has no FFI layer (everything is managed code), so these entry
 * points are what a P/Invoke binding for that method binds instead of running
 *     ConnectivityDecoder.DecodeConnectivity       src/Draco/IO/Mesh/MeshEdgeBreakerDecoder.cs:25-134
 *     ConnectivityDecoder.DecodeAttributes         src/Draco/IO/ConnectivityDecoder.cs:16-44
 * in-process.  The C# binding is in draco-sharp_amd/csharp/ and INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, blittable structs, no callbacks into
 * the caller, library-owned result memory with explicit release.  Every call
 * returns a dsa_status; the per-mesh status of a batch mirrors the exception
 * the C# would raise for that stream (InvalidDataException /
 * NotImplementedException, src/Draco/IO/Extensions/Assertions.cs:5-24,
 * src/Draco/IO/DracoDecoder.cs:49,70,87-97).  One bad stream never poisons the
 * rest of the batch.
 *
 * Threading: a context is bound to one GPU; its calls that queue work (dsa_batch_create, _decode, _download,
 * dsa_encode_batch) come from one host thread at a time; different contexts (one per GPU) are independent.
 * dsa_batch_wait and dsa_batch_free of a batch may run on another thread than the one that is creating or decoding
 * the context's NEXT batch (the pool does this: a consumer frees job k while the workers decode job k+1): the
 * context's caches of arenas, pinned mirrors and descriptor zones, its count of live batches and the turn of its
 * staging buffers and stream sets are behind a lock.  One batch is still used by one thread at a time.
 */
#ifndef DRACO_MI355X_H_
#define DRACO_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSA_ABI_VERSION 4
#define DSA_MAX_ATTRIBUTES 16  /* attributes per mesh handled by the device path; more: DSA_ERR_NOT_IMPLEMENTED */
#define DSA_NUM_STAGES 8

typedef enum dsa_status {
  DSA_OK = 0,
  DSA_ERR_INVALID_DATA = 1,     /* -> System.IO.InvalidDataException */
  DSA_ERR_NOT_IMPLEMENTED = 2,  /* -> System.NotImplementedException (stream feature outside the device path: see INTEGRATION.md section 5) */
  DSA_ERR_INVALID_ARGUMENT = 3, /* -> System.ArgumentException */
  DSA_ERR_DEVICE = 4,           /* HIP runtime failure; see dsa_last_error */
  DSA_ERR_OUT_OF_MEMORY = 5
} dsa_status;

typedef struct dsa_context dsa_context;
typedef struct dsa_batch dsa_batch;

/* DracoHeader (src/Draco/DracoHeader.cs:5-23) + Mesh/PointCloud counts
 * (src/Draco/IO/Mesh/Mesh.cs:15-69, src/Draco/IO/PointCloud/PointCloud.cs:11-133). */
typedef struct dsa_mesh_info {
  int32_t status;            /* dsa_status of this stream */
  int32_t detail;            /* internal site code of the first failing check (diagnostics): 1xx header / section walk
                              * (k_locate), 2xx Edgebreaker connectivity, 3xx traversal, 4xx entropy decode, 5xx prediction,
                              * 6xx general path (valence, seams, corner attributes, sequential meshes) */
  uint8_t major_version, minor_version, encoder_type, encoder_method;
  uint16_t flags;
  uint16_t decode_path;      /* which kernels decoded the stream: 0 the wave-per-mesh kernels, 1 the general path (predictive traversal,
                              * attribute seams, corner attributes, sequential meshes, the schemes of INTEGRATION.md section 5),
                              * 2 the general path at the second attempt (what the fast kernels found behind the symbol streams) */
  uint32_t num_faces;
  uint32_t num_points;
  uint32_t num_attributes;
  uint64_t drc_bytes;        /* length of the compressed stream */
} dsa_mesh_info;

/* PointAttribute / GeometryAttribute (src/Draco/IO/Attributes/PointAttribute.cs:5-63,
 * GeometryAttribute.cs:8-67) + AttributeTransformData. */
typedef struct dsa_attribute_info {
  int32_t attribute_type;    /* GeometryAttributeType: 0 position, 1 normal, 2 color, 3 texcoord, 4 generic */
  int32_t data_type;         /* DataType enum (src/Draco/IO/Enums/DataType.cs) */
  int32_t num_components;
  int32_t normalized;
  uint32_t unique_id;
  uint32_t num_entries;      /* UniqueEntriesCount: values in traversal order */
  uint32_t byte_stride;      /* packed AoS: size(data_type) * num_components */
  int32_t decoder_type;      /* SequentialAttributeEncoderType: 0 generic, 1 integer, 2 quantization, 3 normals */
  int32_t prediction_method; /* PredictionSchemeMethod */
  int32_t prediction_transform;
  int32_t quantization_bits; /* quantization / octahedron transform parameter */
  float range;               /* quantization transform */
  float min_values[4];
} dsa_attribute_info;

int dsa_abi_version(void);
int dsa_device_count(void);

/* Creates a context on `device`.  `stream` is a hipStream_t to run on (NULL =
 * the context creates its own non-blocking stream). */
dsa_status dsa_context_create(int device, void *stream, dsa_context **out);
void dsa_context_destroy(dsa_context *ctx);
/* Message of the last failing call on this context (valid until the next call). */
const char *dsa_last_error(const dsa_context *ctx);

/* Builds a batch from n host-resident .drc streams: parses the fixed headers to
 * size the device arena, takes it from the context's cache (or allocates it) and
 * queues the upload of the compressed bytes.  The streams are copied into pinned
 * staging memory before the call returns -- the caller's buffers are not
 * referenced afterwards -- and travel in one DMA on the context's upload stream,
 * beside the kernels of a batch decoded meanwhile; dsa_batch_decode orders its
 * kernels behind that transfer. */
dsa_status dsa_batch_create(dsa_context *ctx, uint32_t n, const uint8_t *const *streams, const size_t *lengths,
                            dsa_batch **out);
/* Same, for streams stored back to back in one blob: stream i is
 * blob[offsets[i] .. offsets[i+1]). */
dsa_status dsa_batch_create_packed(dsa_context *ctx, uint32_t n, const uint8_t *blob, const uint64_t *offsets,
                                   dsa_batch **out);
/* Enqueues the device-resident decode of the whole batch (compressed bytes in
 * HBM -> faces, attribute values and point maps in HBM).  Asynchronous.  A context
 * created without a stream of the caller's owns two sets of streams and uses them in
 * turn, so that a decode queued while the previous batch is still running starts beside
 * that batch's tail instead of behind it (two device-resident batches in flight); a
 * context created on a caller's stream runs every decode in that stream's order.
 * Decoding the same batch again waits for its previous decode. */
dsa_status dsa_batch_decode(dsa_batch *batch);
/* Waits for the decode (and for a download queued with dsa_batch_download) and collects
 * the per-mesh results.  Waits for this batch only: another batch of the same context may
 * be queued behind it (upload of batch k+1 beside the kernels of batch k beside the
 * download of batch k-1 is the intended use; a context keeps up to three arenas and
 * pinned mirrors of freed batches for that). */
dsa_status dsa_batch_wait(dsa_batch *batch);
void dsa_batch_free(dsa_batch *batch);

uint32_t dsa_batch_size(const dsa_batch *batch);
/* Algorithmic bytes of the decoded batch: compressed
 * bytes read + faces + attribute values + explicit point maps written. */
uint64_t dsa_batch_algorithmic_bytes(const dsa_batch *batch);
/* Device bytes of the batch arena (inputs + outputs + scratch). */
uint64_t dsa_batch_arena_bytes(const dsa_batch *batch);

dsa_status dsa_batch_mesh_info(const dsa_batch *batch, uint32_t mesh, dsa_mesh_info *out);
dsa_status dsa_batch_attribute_info(const dsa_batch *batch, uint32_t mesh, uint32_t attribute, dsa_attribute_info *out);

/* Copy-out (device -> caller memory). */
dsa_status dsa_batch_copy_faces(const dsa_batch *batch, uint32_t mesh, int32_t *dst /* num_faces*3, point ids */);
dsa_status dsa_batch_copy_attribute_values(const dsa_batch *batch, uint32_t mesh, uint32_t attribute, void *dst);
dsa_status dsa_batch_copy_point_map(const dsa_batch *batch, uint32_t mesh, uint32_t attribute, uint32_t *dst /* num_points */);
/* Portable (pre-transform) int32 values, for integer-exactness checks. */
dsa_status dsa_batch_copy_portable_values(const dsa_batch *batch, uint32_t mesh, uint32_t attribute, int32_t *dst);

/* Whole-batch copy-out, what DracoDecoder.Decode's caller needs (src/Draco/IO/DracoDecoder.cs:19-42 returns host
 * objects: Mesh.Faces, PointAttribute buffers, src/Draco/IO/Attributes/PointAttribute.cs:38-63).  The arrays a caller
 * receives -- faces, attribute values and point maps of every mesh -- lie in one block of the batch's arena
 * (dsa_batch_output_bytes long).  dsa_batch_download queues ONE device -> host transfer of that block on the context's
 * download stream, ordered behind the batch's kernels, into `dst` (at least dsa_batch_output_bytes; pinned memory --
 * dsa_host_alloc or dsa_host_register -- for the link's full rate) or, with dst == NULL, into a pinned mirror the library
 * owns (valid until dsa_batch_free or the next dsa_batch_decode of the batch).  Asynchronous: dsa_batch_wait waits for it.
 * dsa_batch_output_layout gives the byte offsets of a mesh's arrays inside the block (array lengths: dsa_batch_mesh_info /
 * dsa_batch_attribute_info).  Meshes that were decoded a second time through the general path
 * live in a second block (block == 1), always mirrored by the library.  After a download the dsa_batch_copy_* calls above
 * are served from the host copy. */
#define DSA_OUTPUT_FACES_U16 1u                 /* dsa_mesh_output.flags: the faces of this mesh are uint16 (compact download, <= 65 536 points) */
typedef struct dsa_mesh_output {
  uint32_t block;                               /* 0: the batch's block (dst / mirror), 1: the block of the re-decoded meshes */
  uint32_t flags;                               /* DSA_OUTPUT_* (0 after a full download) */
  uint64_t faces;                               /* int32[num_faces * 3], or uint16[num_faces * 3] with DSA_OUTPUT_FACES_U16 */
  uint64_t values[DSA_MAX_ATTRIBUTES];          /* attribute a: num_entries * byte_stride bytes */
  uint64_t point_map[DSA_MAX_ATTRIBUTES];       /* attribute a: uint32[num_points]; compact download: attributes decoded in one order
                                                 * share one map (equal offsets), UINT64_MAX: the identity (point i = entry i), not stored */
} dsa_mesh_output;
uint64_t dsa_batch_output_bytes(const dsa_batch *batch);
dsa_status dsa_batch_download(dsa_batch *batch, void *dst, size_t dst_bytes);
/* The same with less on the link (ABI 3): the attribute values as they are, the faces as uint16 wherever a mesh has at most 65 536
 * points, and one point map per DISTINCT map -- the attributes of a mesh that were decoded in one traversal order (all per-vertex
 * attributes; the attributes of one corner-attribute decoder) share theirs, the identity map of a point cloud is not stored.  The
 * packing runs on the device behind the decode; dsa_batch_output_layout then describes the compact host copy (flags, shared /
 * absent maps) and the dsa_batch_copy_* calls widen from it.  Meshes decoded a second time (block 1) keep the full layout.
 * 2.25 -> 1.59 MB for a 64k-triangle mesh with three per-vertex attributes. */
uint64_t dsa_batch_compact_bytes(const dsa_batch *batch);
dsa_status dsa_batch_download_compact(dsa_batch *batch, void *dst, size_t dst_bytes);
const void *dsa_batch_host_output(const dsa_batch *batch, uint32_t block);   /* NULL until dsa_batch_wait has seen the download finish */
dsa_status dsa_batch_output_layout(const dsa_batch *batch, uint32_t mesh, dsa_mesh_output *out);
/* Pinned host memory for download destinations (and for inputs a caller reuses). */
void *dsa_host_alloc(size_t bytes);
void dsa_host_free(void *p);
dsa_status dsa_host_register(void *p, size_t bytes);
dsa_status dsa_host_unregister(void *p);

/* Device pointers of the results, for consumers that stay on the GPU.  Valid
 * until dsa_batch_free. */
const int32_t *dsa_batch_device_faces(const dsa_batch *batch, uint32_t mesh);
const void *dsa_batch_device_attribute_values(const dsa_batch *batch, uint32_t mesh, uint32_t attribute);
const uint32_t *dsa_batch_device_point_map(const dsa_batch *batch, uint32_t mesh, uint32_t attribute);

/* The metadata block of a stream (header flag 0x8000; what Metadata/MetadataDecoder.cs:5-49 reads: per-attribute
 * elements, then the file element), byte for byte.  The decode path skips it structurally; the managed side parses
 * these bytes into DracoMetadata when the caller asks.  dst may be NULL to query *length (0: no metadata). */
dsa_status dsa_batch_copy_metadata(const dsa_batch *batch, uint32_t mesh, uint8_t *dst, size_t dst_bytes, size_t *length);

/* Diagnostics for the parity tests: intermediate products of the path.
 * what: 0 opposite[3F], 1 corner_to_vertex[3F], 2 data_to_corner[entries], 3 vertex_to_data[vertices],
 *       4 uint32[20] clocks recorded by the per-mesh kernels (s_memtime deltas between phases; [13..17] ticks, start and
 *         duration of the connectivity and the traversal wave in s_memrealtime ticks: readable for failed meshes too),
 *       5 uint32[DSA_MAX_ATTRIBUTES][4] per attribute {symbol source, alphabet size, rANS precision bits, rANS payload bytes},
 *       6 the traversal trace of a -DDSA_TRAV_TRACE build. */
dsa_status dsa_batch_copy_debug(const dsa_batch *batch, uint32_t mesh, int what, void *dst, size_t dst_bytes, size_t *written);

/* Per-stage device time of the last dsa_batch_decode, in ms (HIP events on the
 * context's stream; enabled with dsa_context_set_profiling).  names[] receives
 * static strings. */
dsa_status dsa_context_set_profiling(dsa_context *ctx, int enabled);
dsa_status dsa_batch_stage_times(const dsa_batch *batch, float ms[DSA_NUM_STAGES], const char *names[DSA_NUM_STAGES]);
/* Durations of the step's main kernels, each from an event pair of its own on the stream it was launched on (profiling
 * enabled): what a row of `rocprofv3 --kernel-trace --stats` shows for that kernel.  *count receives how many kernels
 * were timed in the last decode; at most `capacity` entries of ms[] / names[] (static strings) are written. */
dsa_status dsa_batch_kernel_times(const dsa_batch *batch, float *ms, const char **names, uint32_t capacity, uint32_t *count);
/* Releases what the context keeps between calls: arenas, pinned mirrors and descriptor zones of freed batches, the
 * encoder's lanes (device buffers + pinned staging).  The library does this itself when an allocation fails. */
dsa_status dsa_context_trim(dsa_context *ctx);
/* What the context found when it checked the assumptions its kernel schedule rests on (static string owned by the context): the
 * register counts of the kernels whose occupancy the late symbol launch of a crowded batch is timed by, and whether that
 * mechanism (k_register_gate) is in use or was left out because the counts of this build no longer add up. */
const char *dsa_context_schedule_note(const dsa_context *ctx);

/* ------------------------------------------------------------------ encode direction
 * Drop-in for DracoEncoder.Encode(BinaryWriter, Config, PointCloud, attributes)   src/Draco/IO/DracoEncoder.cs:22-41
 * on a batch of triangle meshes with per-vertex attributes: Edgebreaker (standard traversal) connectivity, quantisation,
 * prediction, symbol statistics, scheme selection and rANS coding as HIP kernels (BASELINE.json configs[4]); the host
 * checks index ranges before and lays the bytes of each stream out after (batches below 256 meshes let the host
 * threads do connectivity and symbol plans as well: the device's fixed latency exceeds their work).
 * Options mirror the reference's Config (src/Draco/IO/Config.cs): quantisation bits per attribute type, speed
 * (compression_level = 10 - speed), prediction scheme overrides. */
typedef struct dsa_encode_options {
  int32_t position_bits;       /* 1..20, default 11 */
  int32_t texcoord_bits;       /* 1..20, default 10 */
  int32_t normal_bits;         /* 2..20, default 8  */
  int32_t single_connectivity; /* 0: one attributes decoder per attribute (Draco default at speed 5), 1: one for all */
  int32_t symbol_scheme;       /* -1 choose per stream (SymbolEncoding.cs:8-40), 0 tagged, 1 raw */
  int32_t compression_level;   /* 0..10, default 5 */
  int32_t position_prediction; /* PredictionSchemeMethod: 1 parallelogram (default), 0 difference */
  int32_t texcoord_prediction;
} dsa_encode_options;

typedef struct dsa_mesh_input {
  uint32_t num_vertices, num_faces;
  const float *positions;      /* num_vertices * 3 */
  const uint32_t *faces;       /* num_faces * 3 vertex indices; manifold, no isolated vertices */
  const float *normals;        /* num_vertices * 3 or NULL */
  const float *texcoords;      /* num_vertices * 2 or NULL */
  /* ABI 4: one generic attribute of 1 - 4 uint8 components per vertex (vertex colours, material ids ...), coded as an integer
   * attribute (SequentialAttributeEncoderType.Integer, GeometryAttributeType.Generic) with the prediction of the positions'
   * family (difference / parallelogram); NULL / 0: none */
  const uint8_t *generic;      /* num_vertices * generic_components or NULL */
  uint32_t generic_components;
  uint32_t reserved;
} dsa_mesh_input;

typedef struct dsa_encoded dsa_encoded;

void dsa_encode_default_options(dsa_encode_options *options);
/* Encodes n meshes (host-resident inputs, copied) into n .drc streams.  Synchronous.  A mesh that cannot be
 * encoded (non-manifold, isolated vertex, index out of range) fails alone: see dsa_encoded_stream. */
dsa_status dsa_encode_batch(dsa_context *ctx, uint32_t n, const dsa_mesh_input *meshes, const dsa_encode_options *options,
                            dsa_encoded **out);
uint32_t dsa_encoded_size(const dsa_encoded *encoded);
/* Bytes of stream `mesh` (owned by `encoded`, valid until dsa_encoded_free) or that mesh's failure status. */
dsa_status dsa_encoded_stream(const dsa_encoded *encoded, uint32_t mesh, const uint8_t **bytes, size_t *length);
void dsa_encoded_free(dsa_encoded *encoded);

/* ------------------------------------------------------------------------------------------------------------
 * Multi-GPU submit for a host that is one process (the C# host cannot use torch.distributed): a pool owns one
 * context per listed device and decodes a list of independent streams with one worker thread per context.
 * The reference creates everything per call (src/Draco/IO/DracoDecoder.cs:19-42): streams share nothing, so the
 * partition is free -- streams are sorted by compressed length (longest first, the proxy for decode work), cut into
 * chunks of `chunk_meshes`, and the workers pull chunk after chunk from one atomic queue; a chunk is one batch on the
 * worker's context (no collective, no peer traffic).  A device may be listed more than once
 * (several contexts on one GPU).  dsa_pool_decode blocks until every chunk is decoded; results stay on the device
 * that decoded them and are read through the dsa_batch_* accessors of the batch dsa_pool_job_locate names. */
typedef struct dsa_pool dsa_pool;
typedef struct dsa_pool_job dsa_pool_job;

/* chunk_meshes == 0: chosen per job (one chunk per device up to 4096 meshes each, 4096-mesh chunks beyond).  A worker keeps two
 * chunks in flight on its context (upload of the next beside the kernels of the current).  Jobs may outlive the pool object:
 * dsa_pool_destroy with jobs alive takes effect when the last of them is freed.  One dsa_pool_decode at a time per pool. */
dsa_status dsa_pool_create(const int *devices, uint32_t num_devices, uint32_t chunk_meshes, dsa_pool **out);
void dsa_pool_destroy(dsa_pool *pool);
uint32_t dsa_pool_size(const dsa_pool *pool);                 /* number of contexts / worker threads */
const char *dsa_pool_last_error(const dsa_pool *pool);
dsa_status dsa_pool_decode(dsa_pool *pool, uint32_t n, const uint8_t *const *streams, const size_t *lengths, dsa_pool_job **out);
/* Where stream `stream` of the job was decoded: the batch, its index inside it, and the worker (index into the
 * pool's device list) that took its chunk. */
dsa_status dsa_pool_job_locate(const dsa_pool_job *job, uint32_t stream, const dsa_batch **batch, uint32_t *mesh, uint32_t *worker);
uint32_t dsa_pool_job_chunks(const dsa_pool_job *job);
void dsa_pool_job_free(dsa_pool_job *job);
/* The queue order of a job (no GPU needed): order[0..n) = stream indices longest first (ties by index),
 * chunk_begin[0..chunks] = chunk boundaries in `order` (chunk_begin needs n + 1 entries).  Returns the chunk count. */
uint32_t dsa_pool_plan(uint32_t n, const size_t *lengths, uint32_t chunk_meshes, uint32_t *order, uint32_t *chunk_begin);

#ifdef __cplusplus
}
#endif
#endif /* DRACO_MI355X_H_ */
