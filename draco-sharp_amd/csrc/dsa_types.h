// draco-sharp_amd/csrc/dsa_types.h
// Structures shared by the host side of the C-ABI and the HIP kernels.
#pragma once
#include <stdint.h>

#define DSA_MAX_ATT 16       // attributes per mesh on the device path
#define DSA_MAX_ATT_DATA 15  // non-position connectivity data (num_attribute_data)
#define DSA_INVALID 0xFFFFFFFFu

// status codes written by kernels (== dsa_status for 0..2)
#define ST_OK 0
#define ST_INVALID 1
#define ST_NOTIMPL 2
// detail site of a mesh the fast kernels hand back for the general path (dsa_batch_wait decodes it again there)
#define DSA_SITE_RETRY_GENERAL 170

// symbol source of an attribute's value section
#define SRC_TAGGED 0   // tagged rANS scheme      (Entropy/SymbolDecoding.cs:30-50)
#define SRC_RAW 1      // raw rANS scheme         (Entropy/SymbolDecoding.cs:52-67)
#define SRC_FIXED 2    // uncompressed fixed-width ints (SequentialIntegerAttributeDecoder.cs:68-84)
#define SRC_BYTES 3    // generic attribute: raw bytes  (SequentialAttributeDecoder.cs:75-86)

// Per-attribute descriptor, filled by k_locate from the stream.
struct AttrDesc {
  uint8_t att_type, data_type, nc, normalized;
  uint8_t seq_type;        // SequentialAttributeEncoderType 0..3
  int8_t decoder_id;
  int8_t pred_method;      // PredictionSchemeMethod as stored in the stream
  int8_t pred_kind;        // scheme that runs: 0 delta, 1 parallelogram, 2 geometric normal, 3 texture coordinates (portable), 4 constrained multi-parallelogram
                           // (PredictionSchemeDecoderFactory.cs:24-36)
  int8_t pred_transform;   // PredictionSchemeTransformType
  uint8_t nc_portable;
  uint8_t source;          // SRC_*
  uint8_t fixed_bytes;     // SRC_FIXED width
  uint8_t have_scheme;     // prediction scheme instantiated
  uint8_t precision_bits;  // rANS precision of the symbol (raw) or tag (tagged) stream
  uint8_t q_bits;          // quantisation bits / octahedron bits
  uint8_t early_done;      // the symbol wave itself predicted and dequantised the attribute (k_predict / k_finalize of phase 0 skip it)
  uint8_t tags_done;       // tagged scheme: k_tags decoded the tag stream (tags in the output region, their bit total in `table`)
  int8_t corner_data;      // attribute data id + 1 when the attribute's decoder is a corner-attribute decoder (it has a corner table
                           // of its own, cut along the attribute's seams: MeshAttributeCornerTable.cs), else 0
  uint8_t late_located;    // the walk reached this attribute only behind k_seam_tables (it stands behind a corner attribute whose extent
                           // is its entry count): its symbols, bits and prediction are the late launches'
  uint32_t unique_id;
  uint32_t num_symbols;    // alphabet size
  uint32_t off_table;      // stream offset of the first probability-table byte
  uint32_t off_rans;       // rANS payload offset / size
  uint32_t size_rans;
  uint32_t off_bits;       // tagged: byte offset of the value bit section
  uint32_t off_raw;        // SRC_FIXED / SRC_BYTES payload
  int32_t wrap_min, wrap_max;
  int32_t oct_max_q;
  float q_min[4];
  float q_range;
  uint32_t num_entries;
  uint32_t num_distinct;   // raw scheme: symbols of the alphabet with a non-zero frequency (k_locate counts them while it skips the table)
  uint32_t off_flips;      // geometric normal: stream offset of the rABS block of flip bits (probability byte first);
                           // texture coordinates (portable): of the rABS block of orientation bits
  uint32_t num_orient;     // texture coordinates (portable): orientations in that block (MeshPredictionSchemeTexCoordsPortableDecoder.cs:66-85)
  uint32_t off_crease[4], num_crease[4];   // constrained multi-parallelogram (pred_kind 4): the rABS blocks of the crease flags of the four contexts
  uint32_t late_ready;     // a late attribute on the position connectivity, crowded batch: bit 0 its corrections are stored (symbol wave),
                           // bit 1 the order and the operands are (traversal wave) -- set with atomicOr, and the wave that finds the other
                           // bit set predicts the attribute there and then (bit 2: done; k_predict_wrap of phase 1 skips it)
  uint64_t table;          // arena offset of a cumulative table taken from the batch pool (large alphabets), else 0
};

// Per-mesh descriptor, filled by k_locate / k_connectivity / k_traverse.
struct MeshDesc {
  int32_t status;          // ST_*
  int32_t detail;          // site code of the first failing check
  uint8_t major, minor, encoder_type, encoder_method;
  uint16_t flags;
  uint8_t traversal_type;
  uint8_t num_att_data;
  uint32_t num_enc_vertices, num_faces, num_symbols, num_split_symbols, num_splits;
  uint32_t off_splits;     // first topology-split varint
  uint32_t off_split_bits; // source-edge bit section
  uint32_t off_symbols, size_symbols;
  uint32_t off_start_faces;                 // rABS block (prob_zero byte first)
  uint32_t off_seams[DSA_MAX_ATT_DATA];     // rABS blocks
  uint32_t seam_first[DSA_MAX_ATT_DATA];    // k_conn_checks: index of the first set seam bit (DSA_INVALID: none)
  uint32_t off_attributes;                  // num_attributes_decoders byte
  uint32_t num_decoders, num_attributes;
  uint32_t end_pos;
  // results
  uint32_t num_vertices;   // vertices after the reference's isolated-vertex compaction
  uint32_t num_all_vertices;   // corner-table vertices incl. isolated ones
  uint32_t num_points;
  uint32_t num_entries;    // traversal length
  uint32_t general;        // decoded by k_general (dsa_general.h); the fast kernels skip the mesh
  uint32_t gen_act_nv[DSA_MAX_ATT_DATA];   // general path: vertices of every attribute corner table (phase 2 -> phase 3)
  uint32_t gen_seam_pos[DSA_MAX_ATT_DATA]; // general path: stream offset of every attribute seam rABS block (phase 1 -> phase 2)
  uint32_t gen_dec_entries[DSA_MAX_ATT];   // general path: entries of every attributes decoder (phase 3 sequence -> maps -> values)
  uint32_t off_att_values;     // stream offset of the first attribute data section (behind the attribute headers)
  uint8_t dec_first[DSA_MAX_ATT + 4];   // first attribute of every attributes decoder (+ the total behind the last)
  uint32_t resume_pos;         // the walk stopped in front of a tagged symbol stream: where it is taken up again
  uint8_t values_pending, resume_dec, resume_att, pad_resume;
  uint32_t interior_corners;   // 2 x opposite links made by k_connectivity; one seam bit per link and attribute data
  uint32_t linked_corners;     // corners that hold an opposite, counted by k_point_maps (k_seal compares the two)
  // valence traversal (traversal_type 2, MeshEdgeBreakerTraversalValenceDecoder.cs:22-69) on the fast kernels: the six symbol lists,
  // one per valence class of the vertex the decoder stands on -- raw rANS streams over at most 64 symbols each
  uint32_t val_count[6], val_nsym[6], val_off_table[6], val_off_rans[6], val_size_rans[6];
  uint32_t val_lists_done; // bit c: k_valence_lists decoded list c into the face-output region
  uint8_t val_prec[6];
  uint8_t geo_wide;        // k_vertex_positions: some position is not below 2^30 in magnitude (k_predict_geometric then keeps 64-bit edge vectors)
  uint8_t seam_fast;       // the mesh has corner-attribute decoders and the fast kernels take it (k_seam_tables, k_traverse_att)
  // attribute seams on the fast kernels
  uint16_t corner_mask;    // bit d: attribute data d belongs to a corner-attribute decoder
  uint8_t seam_tables_done;    // k_seam_tables has counted the vertices of the attribute tables (seam_nv)
  uint8_t pad_seam;
  uint32_t seam_nv[DSA_MAX_ATT_DATA];      // vertices of attribute data d's corner table (= entries of its decoder), by k_seam_tables
  uint32_t dbg[20];        // diagnostics of the per-mesh kernels (tools/dbg_phases.py, bench.py): s_memtime deltas between phases;
                           // k_connectivity: [13] its s_memtime ticks, [14] its start and [15] its duration in s_memrealtime ticks
                           // (100 MHz); k_traverse: [6] ticks, [16] start, [17] duration: ticks / duration = the shader clock
  AttrDesc att[DSA_MAX_ATT];
};

// Scratch of the general path inside MeshLayout::gen (byte offsets from it).  The same function sizes the
// region on the host and places the arrays in k_general.  F faces, V = cap_vertices, S split events, A attribute
// data, len = stream bytes.  Arrays that the fast kernels own (frec, vrec, vvis, ...) are reused as well; see
// dsa_general.h.
struct GenLayout {
  uint64_t stack;      // u32[F]      active corner stack of the Edgebreaker machine
  uint64_t splits;     // u32[3S]     topology split events (source, split, edge)
  uint64_t active;     // u32[F]      active split corners: decoder symbol id -> corner (only when S > 0)
  uint64_t fvis;       // u8[F]       traversal: face visited
  uint64_t vvis;       // u8[NVmax]   traversal: vertex visited
  uint64_t dfs;        // u32[F+1]    traversal stack
  uint64_t cum;        // u32[cum_entries] cumulative frequencies of the stream being decoded
  uint64_t cum_entries;
  uint64_t pd_next;    // u32[3F]     prediction-degree traversal: the three priority stacks as linked lists over corners
  uint64_t pd_degree;  // u32[NVmax]  prediction-degree traversal: faces a vertex has been seen from
  uint64_t data;       // first per-attribute-data block
  uint64_t data_stride;
  // inside a per-attribute-data block
  uint64_t edge_seam;  // u8[3F]
  uint64_t vert_seam;  // u8[V]
  uint64_t c2v;        // u32[3F]     attribute vertex per corner
  uint64_t v2lm;       // u32[3F]     left-most corner per attribute vertex
  uint64_t d2c;        // u32[NVmax]  entry -> corner
  uint64_t v2d;        // i32[NVmax]  attribute vertex -> entry
  uint64_t pids;       // u32[NVmax]  entry -> point
  uint64_t orient;     // u8[NVmax]   TexCoordsPortable orientations / GeometricNormal flip bits
  uint64_t para;       // u32[3 NVmax] parallelogram operands of every entry (next, prev, opposite entry; next = INVALID: delta)
  uint64_t total;
};
#if defined(__HIPCC__)
__host__ __device__
#endif
inline GenLayout gen_layout(uint64_t F, uint64_t V, uint64_t S, uint64_t A, uint64_t len) {
  GenLayout g;
  const uint64_t C = 3 * F, NV = C > V ? C : V;
  uint64_t cur = 0;
  auto take = [&](uint64_t bytes) { uint64_t at = cur; cur = (cur + bytes + 15) & ~15ull; return at; };
  g.stack = take(4 * F);
  g.splits = take(12 * S);
  g.active = take(S ? 4 * F : 0);
  g.fvis = take(F);
  g.vvis = take(NV);
  g.dfs = take(4 * (F + 1));
  g.cum_entries = (len * 64 < (1ull << 20) ? len * 64 : (1ull << 20)) + 2;
  g.cum = take(4 * g.cum_entries);
  g.pd_next = take(4 * C);
  g.pd_degree = take(4 * NV);
  g.data = cur;
  cur = 0;
  g.edge_seam = take(C);
  g.vert_seam = take(V);
  g.c2v = take(4 * C);
  g.v2lm = take(4 * C);
  g.d2c = take(4 * NV);
  g.v2d = take(4 * NV);
  g.pids = take(4 * NV);
  g.orient = take(NV);
  g.para = take(12 * NV);
  g.data_stride = cur;
  g.total = g.data + A * g.data_stride;
  return g;
}

// Scratch of the fast seam path inside MeshLayout::seam (byte offsets): what a mesh with corner-attribute decoders needs beside
// the position tables.  `eseam` is per mesh; then one block per attribute data.  F faces, NVA = 3F: an attribute vertex per corner
// at most.  rec: the attribute's "virtual mesh" -- face records like MeshLayout::frec whose vertices are the ATTRIBUTE's vertices
// and whose opposites are cut along the attribute's seams, so that traverse_wave, the parallelogram operands and the prediction
// kernels run on it unchanged.
struct SeamLayout {
  uint64_t eseam;      // u32[F]  per face: byte k = for corner k the mask of attribute data for which the edge opposite the corner is a seam
  uint64_t vseam;      // u8[V]   vertex touches a seam of some attribute data or the boundary: its corners are numbered by a walk around it
  uint64_t pbase;      // u32[V]  first point of the vertex
  uint64_t data;       // first per-attribute-data block
  uint64_t data_stride;
  // inside a block
  uint64_t bits;       // u32[ceil(3F/2 / 32) + 4]  the seam bits of the stream, one per interior edge in decoder order (k_conn_checks)
  uint64_t rec;        // 16 B x F (compact) / 32 B x F
  uint64_t vbase;      // u32[V]    first attribute vertex of the position vertex
  uint64_t vflag;      // u8[NVA]   bit0 visited, bit1 on the boundary of the attribute's table
  uint64_t d2c;        // u32[NVA]  entry -> corner
  uint64_t v2d;        // i32[NVA]  attribute vertex -> entry
  uint64_t fvis;       // u8[F]
  uint64_t stack;      // u32[F]    DFS stack of the traversal
  uint64_t para;       // u32[3 NVA] parallelogram operands per entry
  uint64_t orient;     // u32[NVA / 32 + 1]  orientation bits of a texture-coordinate attribute of this decoder (k_flip_bits)
  uint64_t total;
};
#if defined(__HIPCC__)
__host__ __device__
#endif
inline SeamLayout seam_layout(uint64_t F, uint64_t V, uint64_t A, bool compact) {
  SeamLayout g;
  const uint64_t NVA = 3 * F;
  uint64_t cur = 0;
  auto take = [&](uint64_t bytes) { uint64_t at = cur; cur = (cur + bytes + 16 + 255) & ~255ull; return at; };
  g.eseam = take(4 * F);
  g.vseam = take(V);
  g.pbase = take(4 * V);
  g.data = cur;
  cur = 0;
  g.bits = take(4 * ((3 * F / 2 + 31) / 32 + 4));
  g.rec = take((compact ? 16 : 32) * F);
  g.vbase = take(4 * V);
  g.vflag = take(NVA);
  g.d2c = take(4 * NVA);
  g.v2d = take(4 * NVA);
  g.fvis = take(F);
  g.stack = take(4 * F);
  g.para = take(12 * NVA);
  g.orient = take(4 * (NVA / 32 + 1));
  g.data_stride = cur;
  g.total = g.data + A * g.data_stride;
  return g;
}

// What k_texcoords_prepare leaves for the serial chain of k_texcoords, per entry of a TexCoordsPortable attribute: everything of
// MeshPredictionSchemeTexCoordsPortablePredictor.cs:46-150 that depends on the mesh and the positions only.
// What k_multipara_prepare leaves for the chain of k_multipara, per entry of a ConstrainedMultiParallelogram attribute: the (up to
// four) parallelograms the reference finds while it swings around the entry's vertex (MeshPredictionSchemeConstrainedMulti-
// ParallelogramDecoder.cs:46-70) as entry ids {next, prev, opposite}, all decoded before the entry; their number in the top three
// bits of every triple's first word.  The crease flags of the attribute (four bit arrays, one per context, back to back in whole
// words) follow the records of the region.
struct MpPrep { uint32_t id[4][3]; };
#define MP_FOUND_SHIFT 29u
#define MP_ID_MASK 0x1FFFFFFFu

struct TcPrep {
  uint32_t next_id, prev_id;   // entries at Next / Previous of the entry's corner; DSA_INVALID: none, or not decoded before this entry
  int64_t pn_norm2;            // |P(prev) - P(next)|^2
  int64_t cn_dot_pn;           // (P(tip) - P(next)) . (P(prev) - P(next))
  int64_t norm;                // IntSqrt(|C - X|^2 * pn_norm2), X = the foot of the tip on the edge
  double inv;                  // 1 / pn_norm2 as the chain's division wants it (its low 32 bits as a double, reciprocal rounded to nearest)
};

// Compact download (dsa_batch_download_compact): where a mesh's faces and point maps go in the packed block the device makes for
// it -- faces as uint16 when every point id fits, one point map per distinct map (attributes under one connectivity share theirs).
struct CompactMesh {
  uint64_t faces;                 // offset inside the packed block
  uint32_t u16, pad;              // faces stored as uint16
  uint64_t map[DSA_MAX_ATT];      // offset of attribute a's map; equal offsets: shared; ~0: the identity (not stored)
};

// Batch-wide device state.
struct BatchGlobals {
  uint64_t pool;                    // arena offset of the table pool
  uint64_t pool_bytes;
  unsigned long long pool_cursor;   // bump allocator, reset before every decode
  unsigned long long pad;
};

// Host-computed placement of one mesh inside the batch arena (byte offsets from
// the arena base) and the capacities the kernels must respect.
struct MeshLayout {
  uint64_t stream;         // compressed bytes (16-byte aligned)
  uint32_t stream_len;
  uint32_t cap_faces;      // F from the header
  uint32_t cap_vertices;   // num_encoded_vertices + num_split_symbols
  uint32_t cap_attributes;
  uint64_t frec;           // face records (dsa_kernels.h, Rec<>): 32 B {v0, v1, v2, 0, o0, o1, o2, 0} or, rec_compact, 16 B of 21-bit fields; corners are quad coded (4*face + k)
  uint64_t vrec;           // uint2[cap_vertices]: .x left-most corner of the vertex, .y vertex at Previous(left-most corner)
  uint64_t d2c;            // u32[cap_vertices]
  uint64_t v2d;            // i32[cap_vertices]
  uint64_t fvis, vvis;     // u8[F], u8[cap_vertices] (vertex flags: bit0 visited, bit1 on boundary)
  uint64_t fstamp, vstamp; // u32[F], u32[cap_vertices]: (run id, position) stamps of the speculative traversal runs
  uint64_t splits;         // u32[4*cap_splits]: events (source, split|edge<<31), then active pairs (decoder symbol id, corner)
  uint32_t cap_splits;
  uint32_t rec_compact;    // 16-byte face records: every corner and vertex id of the mesh is below 2^20
  uint64_t vrank;          // u32[cap_vertices] vertex -> point id (per-attribute connectivity layout)
  uint64_t para;           // u32[3*cap_vertices] parallelogram operand entries per entry
  uint64_t faces;          // i32[3F] output (DFS stack scratch until k_finalize)
  uint64_t work[DSA_MAX_ATT];   // i32[cap_vertices*nc_portable] symbols -> corrections -> portable values
  uint64_t out[DSA_MAX_ATT];    // attribute values, final format
  uint64_t map[DSA_MAX_ATT];    // u32[cap_vertices] point -> entry
  uint32_t work_cap[DSA_MAX_ATT];  // capacity in int32 elements
  uint32_t out_cap[DSA_MAX_ATT];   // capacity in bytes
  // General path (valence traversal, attribute seams, corner attributes; dsa_general.h): scratch region and the
  // capacities that differ from the vertex-attribute case.  gen_bytes == 0: the mesh takes the fast kernels.
  uint64_t gen;
  uint64_t gen_bytes;
  uint32_t cap_points;     // u32 entries of every map[] (3F when the mesh can have seams, else cap_vertices)
  uint32_t pad2;
  // Fast seam path (corner-attribute decoders on the wave-per-mesh kernels): SeamLayout region; seam_bytes == 0: none
  uint64_t seam;
  uint64_t seam_bytes;
  uint64_t tc[DSA_MAX_ATT];     // TcPrep[entry capacity] of every two-component integer / quantised attribute (it may turn out to be
                                // predicted by TexCoordsPortable: the scheme is written behind the symbol stream), else 0
  uint32_t mp_att;              // bit a: tc[a] is sized for MpPrep records + crease flags (the host parse saw ConstrainedMultiParallelogram
                                // at the head of the first attribute's values -- the one scheme byte it can reach without decoding)
  uint32_t pad_mp;
};
