// draco-sharp_amd/csrc/dsa_locate.h
// The stream walk of k_locate (one lane per mesh): records where every section of a stream starts and what every
// attribute is.  No wave intrinsics: tests/hostcheck compiles it for the host to drive the lane-per-chain kernels'
// per-lane bodies under ASan (test infrastructure; the product library runs it only as the kernel k_locate).
#pragma once
#include "dsa_common.h"

namespace dsa {

#if !defined(__HIPCC__)
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
#endif

// locate_mesh: lane 0 of a wave walks the stream and records where every
// section starts.  Tag streams of the tagged symbol scheme are decoded here
// (their bit section has no length prefix), tags stored as bytes in the
// attribute's output buffer for k_symbols.
#define LOC_MAX_TAGS 64
#define SYM_MAX_LDS 4032   // 63 blocks of 64 cumulative entries searched in LDS by k_symbols
#define LOC_LUT_SLOTS 4096   // slot table of a tag stream (12-bit precision)
#define LOC_LDS_WORDS (LOC_LUT_SLOTS / 4 + 4 * LN_RING_CHUNKS + LN_BLOCK / 4)   // one byte per slot + byte ring + the tags of one block

// One attribute's data section.  Returns 1, 0 after a failed check (status set), or 2 when the attribute's symbols are tagged and
// the tag stream is still to be decoded (tags: LOC_TAGS_LATER): everything up to the tag stream is recorded, the caller comes back
// to the attribute when k_tags has decoded it (a.tags_done) or decodes it here after all (LOC_TAGS_HERE).
#define LOC_TAGS_HERE 0
#define LOC_TAGS_LATER 1
#undef RET
#define RET 0
// A corner attribute (a.corner_data != 0) has one entry per vertex of its own corner table, which only k_seam_tables knows (behind
// the connectivity): its section is walked with the count open (`num_entries` = LOC_COUNT_OPEN).  That works for what carries its
// own length -- a raw-coded symbol stream, the prediction data.  Anything else (raw bytes, tagged symbols, uncompressed integers:
// their extent IS the count) stops the walk in front of the attribute (return value 3); k_locate_resume takes it up behind
// k_seam_tables with the count known, and what it locates then is marked late_located.
#define LOC_COUNT_OPEN 0xFFFFFFFFu
__device__ __forceinline__ int locate_attribute_section(Rd &r, MeshDesc *D, AttrDesc &a, const MeshLayout &L, int ai, uint8_t *arena,
                                        uint32_t *s_cum, uint32_t *s_lut, uint32_t num_entries, BatchGlobals *G, int tags) {
  const uint8_t *s = r.p;
  const bool open = num_entries == LOC_COUNT_OPEN;
  a.num_entries = open ? 0u : num_entries;
  if (open && a.seq_type == 0) return 3;
  if (a.seq_type == 0) {   // generic: raw bytes, SequentialAttributeDecoder.cs:75-86
    a.source = SRC_BYTES;
    a.nc_portable = a.nc;
    a.off_raw = r.pos;
    uint64_t bytes = (uint64_t)data_type_length(a.data_type) * a.nc * num_entries;
    REQUIRE(bytes <= L.out_cap[ai], 140);
    r.skip(bytes);
    REQUIRE(r.ok, 141);
    return 1;
  }
  uint32_t nc = a.seq_type == 3 ? 2u : a.nc;   // normals are (s,t) in portable form
  a.nc_portable = (uint8_t)nc;
  uint64_t num_values = open ? 1 : (uint64_t)num_entries * nc;      // (open: "some", the capacity is checked by k_seam_tables)
  REQUIRE(num_values <= L.work_cap[ai], 142);
  int method = (int8_t)r.u8();
  REQUIRE(r.ok && method >= -2 && method < 7, 143);
  a.pred_method = (int8_t)method;
  a.have_scheme = 0;
  a.pred_transform = -1;
  if (method != -2) {
    int tt = (int8_t)r.u8();
    REQUIRE(r.ok && tt >= -1 && tt < 4, 144);
    a.pred_transform = (int8_t)tt;
    if (a.seq_type == 3) a.have_scheme = (tt == 2 || tt == 3);
    else a.have_scheme = (tt == 1);
  }
  uint32_t compressed = r.u8();
  REQUIRE(r.ok, 145);
  if (compressed > 0) {
    if (num_values > 0) {
      uint32_t scheme = r.u8();
      REQUIRE(r.ok && scheme <= 1, 146);
      if (scheme == 1) {
        a.source = SRC_RAW;
        uint32_t mbl = r.u8();
        REQUIRE(r.ok && mbl >= 1 && mbl <= 18, 147);
        a.precision_bits = (uint8_t)rans_precision_bits(mbl);
        uint64_t ns = r.varint();
        REQUIRE(r.ok && ns >= 1 && ns <= (1u << 20), 148);
        a.num_symbols = (uint32_t)ns;
        a.off_table = r.pos;
        REQUIRE(skip_prob_table(r, a.num_symbols, &a.num_distinct), 149);
        if (a.num_symbols > SYM_MAX_LDS) {   // alphabet too large for the LDS search: cumulative table in global memory
          unsigned long long bytes = ((unsigned long long)a.num_symbols + 2) * 4;
          bytes = (bytes + 15) & ~15ull;
          if (bytes <= L.out_cap[ai]) a.table = L.out[ai];       // the attribute's own output region: free until k_finalize
          else {
            // small mesh, large alphabet: the batch pool; when that is spent the mesh goes to the general path
            // (a cumulative table per mesh), decoded again by dsa_batch_wait -- never a verdict that depends on the
            // rest of the batch
            unsigned long long at = atomicAdd(&G->pool_cursor, bytes);
            if (at + bytes > G->pool_bytes) NOTIMPL(DSA_SITE_RETRY_GENERAL);
            a.table = G->pool + at;
          }
        }
        uint64_t size = r.varint();
        a.off_rans = r.pos;
        r.skip(size);
        REQUIRE(r.ok && size >= 1, 150);
        a.size_rans = (uint32_t)size;
      } else {
        if (open) return 3;
        a.source = SRC_TAGGED;
        a.precision_bits = 12;   // SymbolDecoding.cs:34: tag alphabet is 5 bits wide
        uint64_t ns = r.varint();
        REQUIRE(r.ok && ns >= 1 && ns <= LOC_MAX_TAGS, 151);
        a.num_symbols = (uint32_t)ns;
        a.off_table = r.pos;
        uint64_t total_bits = 0;
        uint32_t worst = 0;
        if (a.tags_done || tags == LOC_TAGS_LATER) {
          // the tag stream is k_tags' (the hand-scheduled register-table decoder on one wave per stream): its place is recorded on
          // the way there, its bit total read on the way back
          uint32_t distinct = 0;
          REQUIRE(skip_prob_table(r, a.num_symbols, &distinct), 152);
          a.num_distinct = distinct;
          uint64_t size = r.varint();
          a.off_rans = r.pos;
          r.skip(size);
          REQUIRE(r.ok && size >= 1, 154);
          a.size_rans = (uint32_t)size;
          a.off_bits = r.pos;
          REQUIRE(num_entries <= L.out_cap[ai], 155);
          if (!a.tags_done) return 2;
          total_bits = a.table & 0x00FFFFFFFFFFFFFFull;
          worst = (uint32_t)(a.table >> 56);
        } else {
          REQUIRE(read_prob_table(r, a.num_symbols, s_cum), 152);
          uint32_t c = 0;
          for (uint32_t i = 0; i < a.num_symbols; ++i) { uint32_t pr = s_cum[i]; REQUIRE(pr <= 4096u - c, 153); s_cum[i] = c; c += pr; }
          s_cum[a.num_symbols] = c;
          REQUIRE(c == 4096, 153);
          uint64_t size = r.varint();
          a.off_rans = r.pos;
          r.skip(size);
          REQUIRE(r.ok && size >= 1, 154);
          a.size_rans = (uint32_t)size;
          a.off_bits = r.pos;
          // decode the tag stream; tags -> out buffer (bytes).  This chain sits in front of everything else of the mesh
          // (the bit section behind it has no length field), so it is kept short: the tag from a byte-per-slot table of the
          // 4096 slots (RAnsDecoder.cs:69-88), its range from the cumulative table, stream bytes from the LDS ring, the
          // tags of a 16-entry block staged in LDS and stored with one 16-byte store at the block's boundary -- nothing
          // inside a block waits for global memory.
          REQUIRE(num_entries <= L.out_cap[ai], 155);
          uint8_t *tags = arena + L.out[ai];
          uint32_t state, off;
          REQUIRE(rans_init(s + a.off_rans, a.size_rans, 16384, &state, &off), 156);
          uint8_t *lut8 = (uint8_t *)s_lut;                   // slot -> tag: 4 KB, so that every mesh of a 4096-mesh batch is resident at once
          for (uint32_t i = 0; i < a.num_symbols; ++i)
            for (uint32_t j = s_cum[i]; j < s_cum[i + 1]; ++j) lut8[j] = (uint8_t)i;
          uint32_t *ring = s_lut + LOC_LUT_SLOTS / 4, *tagbuf = ring + 4 * LN_RING_CHUNKS;
          const uint8_t *ring8 = (const uint8_t *)ring;
          uint8_t *tagbuf8 = (uint8_t *)tagbuf;
          const uint64_t lowest = L.stream & ~15ull;
          const uint64_t last = L.stream + a.off_rans + (off ? off - 1u : 0u);
          const uint64_t base = (last + 16ull) & ~15ull;
          uint32_t loaded = 0;
          for (; loaded < 6; ++loaded) {
            const Chunk c = ln_load_chunk(arena, base, lowest, loaded);
            for (int k = 0; k < 4; ++k) ring[((loaded & (LN_RING_CHUNKS - 1u)) << 2) + k] = c.d[k];
          }
          Chunk in0 = ln_load_chunk(arena, base, lowest, loaded), in1 = ln_load_chunk(arena, base, lowest, loaded + 1);
          bool inflight = true;
          uint32_t q = (uint32_t)(base - 1 - last);
          const uint32_t q_end = q + off;
          while (state < 16384 && q < q_end) { state = (state << 8) | ring8[(q ^ 15u) & 127u]; ++q; }
          for (uint32_t b0 = 0; b0 < num_entries; b0 += LN_BLOCK) {
            const uint32_t cnt = num_entries - b0 < LN_BLOCK ? num_entries - b0 : LN_BLOCK;
            for (uint32_t j = 0; j < cnt; ++j) {
              const uint32_t y0 = ring8[(q ^ 15u) & 127u], y1 = ring8[((q + 1u) ^ 15u) & 127u];
              const uint32_t rem = state & 4095u, sym = lut8[rem];
              const uint32_t cs = s_cum[sym], f = s_cum[sym + 1] - cs;
              state = (state >> 12) * f + rem - cs;
              tagbuf8[j] = (uint8_t)sym;
              worst = sym > worst ? sym : worst;
              total_bits += (uint64_t)sym * nc;
              uint32_t nb = (state < 16384u ? 1u : 0u) + (state < 64u ? 1u : 0u);      // state >= 4 after a step: two bytes reach 2^14
              const uint32_t left = q_end - q;
              nb = nb < left ? nb : left;
              state = (state << (8u * nb)) | (((y0 << 8) | y1) >> (16u - 8u * nb));
              q += nb;
              while (state < 16384 && q < q_end) { state = (state << 8) | ring8[(q ^ 15u) & 127u]; ++q; }
            }
            // block boundary: the chunks requested one boundary ago go into the ring, the block's tags leave, two more chunks are requested
            if (inflight) {
              for (int k = 0; k < 4; ++k) { ring[((loaded & (LN_RING_CHUNKS - 1u)) << 2) + k] = in0.d[k]; ring[(((loaded + 1u) & (LN_RING_CHUNKS - 1u)) << 2) + k] = in1.d[k]; }
              loaded += 2;
            }
            if (cnt == LN_BLOCK) {
#if defined(__HIPCC__)
              *(uint4 *)(tags + b0) = make_uint4(tagbuf[0], tagbuf[1], tagbuf[2], tagbuf[3]);     // out regions are 256-byte aligned
#else
              memcpy(tags + b0, tagbuf8, LN_BLOCK);
#endif
            } else {
              for (uint32_t j = 0; j < cnt; ++j) tags[b0 + j] = tagbuf8[j];
            }
            inflight = loaded + 2u - (q >> 4) <= LN_RING_CHUNKS;
            if (inflight) { in0 = ln_load_chunk(arena, base, lowest, loaded); in1 = ln_load_chunk(arena, base, lowest, loaded + 1); }
          }
        }
        REQUIRE(worst <= 32, 157);
        r.skip((total_bits + 7) >> 3);
        REQUIRE(r.ok, 158);
      }
    } else {
      a.source = SRC_RAW;
      a.num_symbols = 0;
    }
  } else {
    if (open) return 3;
    a.source = SRC_FIXED;
    uint32_t nb = r.u8();
    REQUIRE(r.ok && nb >= 1 && nb <= 4, 159);
    a.fixed_bytes = (uint8_t)nb;
    a.off_raw = r.pos;
    r.skip((uint64_t)nb * num_values);
    REQUIRE(r.ok, 160);
  }
  if (a.have_scheme) {
    // only Difference and Parallelogram run on the device path for now
    // PredictionSchemeDecoderFactory.cs:24-36: without a corner table (point clouds) every method falls back to delta
    // With a corner table the schemes a transform carries are (D-26): wrap -> parallelogram family + texture
    // coordinates, octahedral -> geometric normal only; every other combination is the delta scheme.  Schemes that
    // need the general path's tables send the mesh there: the host decodes it again (site DSA_SITE_RETRY_GENERAL).
    if (D->encoder_type == 0) a.pred_kind = 0;
    else if (a.pred_transform == 1) {
      if (method == 2) NOTIMPL(DSA_SITE_RETRY_GENERAL);
      // ConstrainedMultiParallelogram on the fast kernels where the host parse set its records aside (a mesh whose first attribute
      // shows the scheme: every attribute on the position connectivity with at most four components; located in the first pass)
      // -- k_crease_bits, k_multipara_prepare, k_multipara (dsa_seams.h)
      if (method == 4 && (!((L.mp_att >> ai) & 1u) || L.tc[ai] == 0 || nc > 4 || a.corner_data != 0 || a.late_located != 0)) NOTIMPL(DSA_SITE_RETRY_GENERAL);
      if (!(method == 0 || method == 1 || method == 4 || method == 5 || method == 6)) NOTIMPL(161);
      a.pred_kind = method == 1 ? 1 : (method == 5 ? 3 : (method == 4 ? 4 : 0));
      if (a.pred_kind == 4) {
        // MeshPredictionSchemeConstrainedMultiParallelogramDecoder.cs:110-134 (v2.2: no mode byte): per context the number of crease
        // flags and, if any, an rABS block of them
        uint64_t total = 0;
        for (int k = 0; k < 4; ++k) {
          const uint64_t nf = r.varint();
          REQUIRE(r.ok && nf <= 4ull * L.cap_vertices, 673);
          a.num_crease[k] = (uint32_t)nf; a.off_crease[k] = 0;
          total += nf;
          if (nf > 0) {
            Rabs rd;
            uint32_t endp;
            rd.start(s, L.stream_len, r.pos, &endp);
            REQUIRE(rd.ok, 674);
            a.off_crease[k] = r.pos;
            r.pos = endp;
          }
        }
        REQUIRE(total <= 4ull * L.cap_vertices, 673);             // (a parallelogram of an entry takes a flag: four per entry at most)
      }
      if (a.pred_kind == 3) {
        // MeshPredictionSchemeTexCoordsPortableDecoder.cs:66-85: the orientation count and an rABS block of their deltas, in front of
        // the transform data; the scheme needs two components and the portable positions (three) decoded before it
        REQUIRE(nc == 2, 666);
        int pa = -1;
        for (int k = 0; k < ai; ++k) if (D->att[k].att_type == 0 && D->att[k].seq_type != 0) { pa = k; break; }
        REQUIRE(pa >= 0 && D->att[pa].nc_portable == 3, 667);
        if (D->att[pa].corner_data != 0) NOTIMPL(DSA_SITE_RETRY_GENERAL);      // positions with a corner table of their own: no encoder writes that
        const int32_t num_or = (int32_t)r.u32();
        REQUIRE(r.ok && num_or >= 0, 668);
        Rabs rd;
        uint32_t endp;
        rd.start(s, L.stream_len, r.pos, &endp);
        REQUIRE(rd.ok, 669);
        a.off_flips = r.pos;
        a.num_orient = (uint32_t)num_or;
        r.pos = endp;
        if (a.corner_data != 0) for (int k = 0; k < ai; ++k) if (D->att[k].corner_data == a.corner_data && D->att[k].have_scheme && (D->att[k].pred_kind == 2 || D->att[k].pred_kind == 3)) NOTIMPL(DSA_SITE_RETRY_GENERAL);
      }
    } else {
      // GeometricNormal (MeshPredictionSchemeGeometricNormalDecoder.cs:44-82): predicted from the decoded positions, entry by entry
      // independently (k_predict_geometric); needs the portable positions before it in the same decoder
      a.pred_kind = method == 6 ? 2 : 0;
      // (normals with seams: the fan of an entry ends at the attribute's seams, k_predict_geometric walks the attribute's own records)
    }
    if (a.pred_transform == 1) {           // PredictionSchemeWrapDecodingTransform.cs:69-75
      a.wrap_min = (int32_t)r.u32();
      a.wrap_max = (int32_t)r.u32();
      REQUIRE(r.ok && a.wrap_min <= a.wrap_max, 162);
      int64_t dif = (int64_t)a.wrap_max - (int64_t)a.wrap_min;
      REQUIRE(dif < 0x7FFFFFFF, 163);
    } else {                               // NormalOctahedron(+Canonicalized)DecodingTransform
      int32_t max_q = (int32_t)r.u32();
      if (a.pred_transform == 3) (void)r.u32();
      REQUIRE(r.ok && max_q > 0 && (max_q & 1) == 1, 164);
      int q = 32 - __clz(max_q);
      REQUIRE(q >= 2 && q <= 30, 165);
      a.oct_max_q = max_q;
      if (a.pred_kind == 2) {              // the flip bits, one per entry, behind the transform data (:71-82): an rABS block
        Rabs rd;
        uint32_t endp;
        rd.start(s, L.stream_len, r.pos, &endp);
        REQUIRE(rd.ok, 655);
        a.off_flips = r.pos;
        r.pos = endp;
        int pa = -1;                       // parent = portable positions, SequentialAttributeDecoder.cs:58-73
        for (int k = 0; k < ai; ++k) if (D->att[k].att_type == 0 && D->att[k].seq_type != 0) { pa = k; break; }
        REQUIRE(pa >= 0 && D->att[pa].nc_portable == 3, 656);
        if (D->att[pa].corner_data != 0) NOTIMPL(DSA_SITE_RETRY_GENERAL);     // positions with a corner table of their own: no encoder writes that
        REQUIRE(a.corner_data != 0 || num_entries <= L.cap_vertices, 657);   // the decoded flip bits wait in the vertex-stamp region (corner attributes: their block's, k_flip_bits checks)
        // one bit array per attribute data block: a second attribute of the same corner decoder that wants one goes to the general path
        if (a.corner_data != 0) for (int k = 0; k < ai; ++k) if (D->att[k].corner_data == a.corner_data && D->att[k].have_scheme && (D->att[k].pred_kind == 2 || D->att[k].pred_kind == 3)) NOTIMPL(DSA_SITE_RETRY_GENERAL);
      }
    }
  }
  return 1;
}
#undef RET
#define RET

// A context list of valence-coded connectivity under the TAGGED symbol scheme (Entropy/SymbolDecoding.cs:30-50: a rANS stream of bit
// lengths, then the values as bit fields) -- what encoders write for the lists of small meshes (the reference's house_04 sample).
// Decoded by the walking lane itself, into the place k_valence_lists / the connectivity wave would put it; such lists are short
// (longer ones go to the general path).  `r` stands behind the scheme byte; on success it stands behind the list.
#define LOC_VAL_TAGGED_MAX 16384u
#undef RET
#define RET false
__device__ inline bool locate_tagged_valence_list(Rd &r, MeshDesc *D, uint32_t count, uint32_t *out) {
  uint32_t cum[LOC_MAX_TAGS + 1];
  const uint64_t ns = r.varint();
  REQUIRE(r.ok && ns >= 1 && ns <= LOC_MAX_TAGS, 151);
  REQUIRE(read_prob_table(r, (uint32_t)ns, cum), 152);
  uint32_t c = 0;
  for (uint32_t i = 0; i < (uint32_t)ns; ++i) { const uint32_t pr = cum[i]; REQUIRE(pr <= 4096u - c, 153); cum[i] = c; c += pr; }
  cum[ns] = c;
  REQUIRE(c == 4096, 153);
  const uint64_t size = r.varint();
  REQUIRE(r.ok && size >= 1 && size <= (uint64_t)(r.n - r.pos), 154);
  const uint8_t *buf = r.p + r.pos;
  r.skip(size);
  uint32_t state, off;
  REQUIRE(rans_init(buf, (uint32_t)size, 16384, &state, &off), 156);
  const uint8_t *bits = r.p + r.pos;
  const uint32_t nbytes = r.n - r.pos;
  uint64_t bitpos = 0;
  for (uint32_t i = 0; i < count; ++i) {
    while (state < 16384 && off > 0) state = state * 256 + buf[--off];
    const uint32_t rem = state & 4095u;
    uint32_t tag = 0;
    while (tag + 1 < (uint32_t)ns && cum[tag + 1] <= rem) ++tag;
    state = (state >> 12) * (cum[tag + 1] - cum[tag]) + rem - cum[tag];
    REQUIRE(tag <= 32, 157);
    out[i] = tag ? read_bits(bits, nbytes, bitpos, tag) : 0u;
    bitpos += tag;
  }
  r.skip((bitpos + 7) >> 3);
  REQUIRE(r.ok, 158);
  return true;
}
#undef RET
#define RET

// First half of the walk: header, metadata, connectivity sections (up to the attribute section).
__device__ inline void locate_mesh(uint8_t *arena, const MeshLayout &L, MeshDesc *D) {
  const uint8_t *s = arena + L.stream;
  Rd r(s, L.stream_len, 0);
  // header, DracoDecoder.cs:44-64
  REQUIRE(L.stream_len >= 11, 100);
  REQUIRE(s[0] == 'D' && s[1] == 'R' && s[2] == 'A' && s[3] == 'C' && s[4] == 'O', 101);
  r.pos = 5;
  D->major = (uint8_t)r.u8(); D->minor = (uint8_t)r.u8();
  D->encoder_type = (uint8_t)r.u8(); D->encoder_method = (uint8_t)r.u8();
  D->flags = (uint16_t)r.u16();
  REQUIRE(D->major == 2 && D->minor == 2, 102);
  if (D->flags & 0x8000) {   // metadata is skipped structurally (Metadata/MetadataDecoder.cs:5-49)
    uint32_t natt = (uint32_t)r.varint();
    uint32_t pending[16];
    int depth = 0;
    uint32_t elements_left = natt + 1;   // per-attribute elements (each preceded by an id) then the file element
    bool first_level_ids = true;
    (void)first_level_ids;
    for (uint32_t e = 0; e < elements_left && r.ok; ++e) {
      if (e < natt) (void)r.varint();
      // one element, iteratively
      depth = 0;
      pending[0] = 1;
      bool at_key = false;
      while (depth >= 0 && r.ok) {
        if (pending[depth] == 0) { --depth; continue; }
        --pending[depth];
        if (at_key || depth > 0) { uint32_t ks = r.u8(); r.skip(ks); }
        uint32_t ne = (uint32_t)r.varint();
        for (uint32_t i = 0; i < ne && r.ok; ++i) { uint32_t ks = r.u8(); r.skip(ks); uint64_t vs = r.varint(); r.skip(vs); }
        uint32_t nsub = (uint32_t)r.varint();
        if (nsub) { REQUIRE(depth < 15, 103); pending[++depth] = nsub; }
      }
    }
    REQUIRE(r.ok, 104);
  }
  REQUIRE(D->encoder_type <= 1, 105);
  if (L.gen_bytes != 0) {               // the host sized this mesh for the general path (dsa_general.h): k_general parses the rest
    REQUIRE(D->encoder_type == 1 && D->encoder_method <= 1, 107);
    D->general = 1;
    D->end_pos = r.pos;
    return;
  }
  const bool point_cloud = D->encoder_type == 0;
  uint32_t nad = 0;
  if (point_cloud) {
    // Sequential point cloud (the reference stops at DracoDecoder.cs:70; layout of the upstream format):
    // int32 point count, then the attribute section with a linear sequencer (entry i = point i).
    REQUIRE(D->encoder_method <= 1, 107);
    if (D->encoder_method != 0) NOTIMPL(106);               // kd-tree point clouds
    const uint32_t np = r.u32();
    REQUIRE(r.ok && np <= 0x7FFFFFFFu && np == L.cap_vertices, 116);
    D->num_enc_vertices = np; D->num_faces = 0; D->num_att_data = 0;
    D->num_vertices = np; D->num_all_vertices = np; D->num_points = np; D->num_entries = np;
  } else {
    REQUIRE(D->encoder_method <= 1, 107);
    if (D->encoder_method == 0) NOTIMPL(108);                 // sequential mesh
    D->traversal_type = (uint8_t)r.u8();
    REQUIRE(r.ok && D->traversal_type <= 2, 109);
    if (D->traversal_type == 1) NOTIMPL(110);                 // predictive traversal: general path (the host parse routes it there)
    // MeshEdgeBreakerDecoder.cs:35-56
    uint64_t nv = r.varint(), nf = r.varint();
    REQUIRE(r.ok && nf <= 0x7FFFFFFFu / 3 && nv <= nf * 3, 111);
    uint64_t min_face_edges = 3 * nf / 2, max_vertex_edges = nv * (nv - 1) / 2;
    REQUIRE(max_vertex_edges >= min_face_edges, 112);
    nad = r.u8();
    uint64_t nsym = r.varint();
    REQUIRE(r.ok && nf >= nsym && nf <= nsym + nsym / 3, 113);
    uint64_t nsplit_sym = r.varint();
    REQUIRE(r.ok && nsplit_sym <= nsym, 114);
    if (nad > DSA_MAX_ATT_DATA) NOTIMPL(115);
    D->num_enc_vertices = (uint32_t)nv; D->num_faces = (uint32_t)nf; D->num_att_data = (uint8_t)nad;
    D->num_symbols = (uint32_t)nsym; D->num_split_symbols = (uint32_t)nsplit_sym;
    REQUIRE(nf == L.cap_faces && nv + nsplit_sym == L.cap_vertices, 116);   // host sizing must agree
    // topology splits, MeshEdgeBreakerDecoder.cs:136-193
    uint64_t nsplits = r.varint();
    REQUIRE(r.ok && nsplits <= nf && nsplits <= L.cap_splits, 117);
    D->num_splits = (uint32_t)nsplits;
    D->off_splits = r.pos;
    for (uint64_t i = 0; i < 2 * nsplits; ++i) (void)r.varint();
    D->off_split_bits = r.pos;
    r.skip((nsplits + 7) >> 3);
    REQUIRE(r.ok, 118);
    // traversal sections, MeshEdgeBreakerTraversalDecoder.cs:27-61 (symbol section is `size` bytes, D-3); valence traversal has
    // no explicit symbol section (MeshEdgeBreakerTraversalValenceDecoder.cs:22-35)
    if (D->traversal_type == 0) {
      uint64_t sym_size = r.varint();
      D->off_symbols = r.pos;
      r.skip(sym_size);
      REQUIRE(r.ok, 119);
      D->size_symbols = (uint32_t)sym_size;
    }
    D->off_start_faces = r.pos;
    { (void)r.u8(); uint64_t sz = r.varint(); r.skip(sz); REQUIRE(r.ok && sz >= 1, 120); }
    for (uint32_t i = 0; i < nad; ++i) {
      D->off_seams[i] = r.pos;
      (void)r.u8();
      uint64_t sz = r.varint();
      r.skip(sz);
      REQUIRE(r.ok && sz >= 1, 121);
    }
    if (D->traversal_type == 2) {
      // the six context lists, :36-69: count, then a DecodeSymbols block of one component.  The connectivity wave decodes raw
      // streams over small alphabets itself; anything else (tagged scheme, an alphabet no encoder writes) goes to the general path
      if (!L.rec_compact) NOTIMPL(DSA_SITE_RETRY_GENERAL);
      uint64_t total = 0;
      for (int c = 0; c < 6; ++c) {
        const uint64_t num = r.varint();
        REQUIRE(r.ok && num <= nf && total + num <= nf, 641);
        D->val_count[c] = (uint32_t)num;
        total += num;
        if (num == 0) continue;
        const uint32_t scheme = r.u8();
        REQUIRE(r.ok && scheme <= 1, 146);
        if (scheme != 1) {
          if (num > LOC_VAL_TAGGED_MAX) NOTIMPL(DSA_SITE_RETRY_GENERAL);
          if (!locate_tagged_valence_list(r, D, (uint32_t)num, (uint32_t *)(arena + L.faces) + (total - num))) return;
          D->val_prec[c] = 0;                      // (not a stream for k_valence_lists)
          D->val_lists_done |= 1u << c;
          continue;
        }
        const uint32_t mbl = r.u8();
        REQUIRE(r.ok && mbl >= 1 && mbl <= 18, 147);
        D->val_prec[c] = (uint8_t)rans_precision_bits(mbl);
        const uint64_t ns = r.varint();
        REQUIRE(r.ok && ns >= 1 && ns <= (1u << 20), 148);
        if (ns > 64) NOTIMPL(DSA_SITE_RETRY_GENERAL);
        D->val_nsym[c] = (uint32_t)ns;
        D->val_off_table[c] = r.pos;
        uint32_t distinct = 0;
        REQUIRE(skip_prob_table(r, (uint32_t)ns, &distinct), 149);
        const uint64_t size = r.varint();
        D->val_off_rans[c] = r.pos;
        r.skip(size);
        REQUIRE(r.ok && size >= 1, 150);
        D->val_size_rans[c] = (uint32_t)size;
      }
    }
  }
  D->off_attributes = r.pos;
}
__device__ inline void locate_attribute_headers(uint8_t *arena, const MeshLayout &L, MeshDesc *D);
#define LOC_WHOLE 0        // every data section, tag streams decoded on the way (host; the device when it has no choice)
#define LOC_UNTIL_TAGS 1   // from the first data section up to the first tagged symbol stream (k_locate)
#define LOC_RESUME 2       // from the attribute the walk stopped at (k_locate_resume): its tags are k_tags' by now, as a rule
__device__ inline void locate_attribute_values(uint8_t *arena, const MeshLayout &L, MeshDesc *D, BatchGlobals *G, uint32_t *s_cum, uint32_t *s_lut, int mode);
// The whole walk of one stream.  On the device it comes in pieces: the bit section behind a tag stream has no length field, so
// what follows a tagged attribute is only found by decoding its tags -- a serial chain of 33 000 steps for a bench mesh that
// one lane of k_locate used to walk in front of everything (23 ms for the bench batch with tagged symbols).  Now k_locate stops in
// front of a tag stream, k_tags decodes it with the register-table decoder (one wave per stream, nine instructions per tag), and
// k_locate_resume takes the walk up behind it -- up to the next tag stream; the host queues as many rounds as a mesh has attributes.
__device__ inline void locate_all(uint8_t *arena, const MeshLayout &L, MeshDesc *D, BatchGlobals *G, uint32_t *s_cum, uint32_t *s_lut) {
  locate_mesh(arena, L, D);
  if (D->status != ST_OK || D->general) return;
  locate_attribute_headers(arena, L, D);
  if (D->status != ST_OK) return;
  locate_attribute_values(arena, L, D, G, s_cum, s_lut, LOC_WHOLE);
}

// Second part of the walk: the headers of the attribute section (ConnectivityDecoder.cs:16-44) from D->off_attributes on.
__device__ inline void locate_attribute_headers(uint8_t *arena, const MeshLayout &L, MeshDesc *D) {
  const uint8_t *s = arena + L.stream;
  Rd r(s, L.stream_len, D->off_attributes);
  const bool point_cloud = D->encoder_type == 0;
  const uint32_t nad = D->num_att_data;
  uint32_t ndec = r.u8();
  REQUIRE(r.ok, 122);
  if (ndec > DSA_MAX_ATT) NOTIMPL(122);
  D->num_decoders = ndec;
  int att_data_of[DSA_MAX_ATT];
  bool pos_seen = false;
  uint32_t data_seen = 0;
  for (uint32_t i = 0; i < ndec && !point_cloud; ++i) {           // MeshEdgeBreakerDecoder.cs:640-708
    int att_data_id = (int8_t)r.u8();
    uint32_t element_type = r.u8();
    uint32_t traversal_method = r.u8();
    REQUIRE(r.ok && traversal_method < 2, 123);
    if (att_data_id >= 0) {
      REQUIRE((uint32_t)att_data_id < nad && !((data_seen >> att_data_id) & 1), 124);
      data_seen |= 1u << att_data_id;
    } else {
      REQUIRE(!pos_seen, 125);
      pos_seen = true;
    }
    // Any other element type than "vertex" is a corner attribute (MeshEdgeBreakerDecoder.cs:666-701): the attribute's own corner
    // table, cut along its seams.  The host sized the seam scratch for such a mesh (MeshLayout::seam).
    if (element_type != 0) {
      REQUIRE(att_data_id >= 0 && traversal_method == 0, 126);
      if (L.seam_bytes == 0 || att_data_id >= 7) NOTIMPL(127);
      D->corner_mask |= (uint16_t)(1u << att_data_id);
      D->seam_fast = 1;
    }
    if (traversal_method != 0) NOTIMPL(DSA_SITE_RETRY_GENERAL);                                     // prediction-degree traversal: general path (the host parse routes it there)
    att_data_of[i] = element_type != 0 ? att_data_id : -1;
  }
  uint32_t natt = 0;
  uint32_t first_att[DSA_MAX_ATT + 1];
  for (uint32_t i = 0; i < ndec; ++i) {           // AttributesDecoder.cs:19-63 + controller :16-27
    first_att[i] = natt;
    uint64_t k = r.varint();
    REQUIRE(r.ok, 129);
    if (natt + k > DSA_MAX_ATT) NOTIMPL(129);
    REQUIRE(natt + k <= L.cap_attributes, 129);
    for (uint32_t j = 0; j < (uint32_t)k; ++j) {
      AttrDesc &a = D->att[natt + j];
      a.att_type = (uint8_t)r.u8(); a.data_type = (uint8_t)r.u8(); a.nc = (uint8_t)r.u8(); a.normalized = r.u8() != 0;
      REQUIRE(r.ok && a.att_type < 5 && a.data_type != 0 && a.data_type < 12 && a.nc != 0, 130);
      a.unique_id = (uint32_t)r.varint();
      a.decoder_id = (int8_t)i;
      a.corner_data = (int8_t)((!point_cloud && i < ndec && att_data_of[i] >= 0) ? att_data_of[i] + 1 : 0);
    }
    for (uint32_t j = 0; j < (uint32_t)k; ++j) {
      AttrDesc &a = D->att[natt + j];
      a.seq_type = (uint8_t)r.u8();
      REQUIRE(r.ok && a.seq_type <= 3, 131);
      if (a.seq_type == 2) REQUIRE(a.data_type == 9 && a.nc <= 4, 132);
      if (a.seq_type == 3) REQUIRE(a.data_type == 9 && a.nc == 3, 133);
      if (a.seq_type == 1) { uint32_t w = data_type_length(a.data_type); REQUIRE(w == 1 || w == 2 || w == 4, 134); }
    }
    natt += (uint32_t)k;
  }
  first_att[ndec] = natt;
  D->num_attributes = natt;
  for (uint32_t i = 0; i <= ndec; ++i) D->dec_first[i] = (uint8_t)first_att[i];
  D->off_att_values = r.pos;
}

// Third part: the data sections of the attributes (their symbol streams, prediction data, quantisation parameters).
__device__ inline void locate_attribute_values(uint8_t *arena, const MeshLayout &L, MeshDesc *D, BatchGlobals *G, uint32_t *s_cum, uint32_t *s_lut, int mode) {
  const uint8_t *s = arena + L.stream;
  Rd r(s, L.stream_len, D->off_att_values);
  const uint32_t ndec = D->num_decoders;
  const uint8_t *first_att = D->dec_first;
  uint32_t i0 = 0, a0 = 0;
  if (mode == LOC_RESUME) {
    if (!D->values_pending) return;
    i0 = D->resume_dec; a0 = D->resume_att; r.pos = D->resume_pos;
    D->values_pending = 0;
  }
  // every vertex attribute of a valid stream carries one entry per encoded vertex
  uint32_t num_entries = D->num_enc_vertices;
  for (uint32_t i = i0; i < ndec; ++i) {           // AttributesDecoder.cs:65-70
    for (uint32_t ai = (i == i0 && a0 > first_att[i]) ? a0 : first_att[i]; ai < first_att[i + 1]; ++ai) {
      const uint32_t at = r.pos;
      // a tag stream ahead is left to k_tags -- unless this IS the attribute the walk stopped at and k_tags did not take it
      // (a stream it is not made for): then its tags are decoded here
      const bool again = mode == LOC_RESUME && i == i0 && ai == a0;
      AttrDesc &A = D->att[ai];
      const uint32_t count = !A.corner_data ? num_entries : (D->seam_tables_done ? D->seam_nv[(uint32_t)A.corner_data - 1u] : LOC_COUNT_OPEN);
      if (D->seam_tables_done) A.late_located = 1;
      const int rc = locate_attribute_section(r, D, A, L, (int)ai, arena, s_cum, s_lut, count, G,
                                              mode == LOC_WHOLE || again ? LOC_TAGS_HERE : LOC_TAGS_LATER);
      if (rc == 0) return;
      if (rc == 2) { D->resume_dec = (uint8_t)i; D->resume_att = (uint8_t)ai; D->resume_pos = at; D->values_pending = 1; return; }
      if (rc == 3) {     // behind k_seam_tables; everything from here on is marked now, so that no early kernel mistakes a half-written descriptor for its own
        for (uint32_t k = ai; k < D->num_attributes; ++k) D->att[k].late_located = 1;
        D->resume_dec = (uint8_t)i; D->resume_att = (uint8_t)ai; D->resume_pos = at; D->values_pending = 2;
        return;
      }
    }
    for (uint32_t ai = first_att[i]; ai < first_att[i + 1]; ++ai) {
      AttrDesc &a = D->att[ai];
      if (a.seq_type == 2) {                      // AttributeQuantizationTransform.cs:110-121
        for (uint32_t c = 0; c < a.nc; ++c) a.q_min[c] = r.f32();
        a.q_range = r.f32();
        a.q_bits = (uint8_t)r.u8();
        REQUIRE(r.ok && a.q_bits >= 1 && a.q_bits <= 30, 135);
      } else if (a.seq_type == 3) {               // AttributeOctahedronTransform.cs:39-42 (D-5)
        a.q_bits = (uint8_t)r.u8();
        REQUIRE(r.ok && a.q_bits >= 2 && a.q_bits <= 30, 136);
      }
    }
  }
  D->end_pos = r.pos;
}

}  // namespace dsa
