// draco-sharp_amd/csrc/dsa_seams.h
// -----------------------------------------------------------------------------
// Attribute seams on the wave-per-mesh kernels: corner-attribute decoders -- attributes with a
// corner table of their own, the position table cut along the attribute's seams -- without the serial general path.
//   MeshEdgeBreakerDecoder.cs:502-535   seam bits -> seam edges              k_conn_checks (the bits), k_seam_tables (the edges)
//   MeshAttributeCornerTable.cs:19-155  attribute vertices per corner        k_seam_tables
//   MeshEdgeBreakerDecoder.cs:537-638   points per corner                    k_seam_tables
//   DepthFirstTraverser.cs:9-99 on the attribute's table                     k_traverse_att (= traverse_wave on the "virtual mesh")
//   MeshTraversalSequencer.cs:33-50     point -> entry maps                  k_seam_maps
//   MeshPredictionSchemeTexCoordsPortablePredictor.cs:46-150                 k_texcoords_prepare + k_texcoords (both kinds of mesh)
// The idea: what makes an attribute's corner table different from the position table is (a) which opposites exist and (b) what the
// vertex of a corner is.  k_seam_tables writes both into a second set of face records per seamed attribute; every kernel that
// works from face records (traversal, parallelogram operands, wrap prediction) then runs on that "virtual mesh" unchanged.
// Numbering attribute vertices and points is a prefix sum over the position vertices with a walk around the few vertices a seam or
// the boundary touches; every other vertex owns exactly one of each.
// -----------------------------------------------------------------------------
#pragma once

namespace dsa {

template <bool CP>
__device__ __forceinline__ uint32_t pos_swing_right(const uint32_t *frec, uint32_t c) {
  const uint32_t o = Rec<CP>::get_o_plain(frec, qprev(c));
  return o == DSA_INVALID ? DSA_INVALID : qprev(o);
}
// field k of the vertex word of a compact record: all ones -> id (the records of a virtual mesh start with every vertex field
// "unset"; the fields of one word are set by different lanes)
__device__ __forceinline__ void vrec_set_compact(uint32_t *rec, uint32_t c, uint32_t id) {
  const uint32_t sh = 21u * (c & 3u);
  const unsigned long long M = 0x1FFFFFull;
  atomicAnd((unsigned long long *)rec + (size_t)(c >> 2) * 2, ((unsigned long long)(id & 0x1FFFFFu) << sh) | ~(M << sh));
}

#define SM_FAIL(code_, site_) { if (lane == 0) fail(D, (code_), (site_)); return; }
#define SM_SYNC() { WAIT_VM0(); __threadfence_block(); }
#define SM_U 4u      // chunks of 64 elements a pass keeps in flight

// One wave per mesh, behind the connectivity (k_connectivity) and the seam bits (k_conn_checks).
template <bool CP>
__device__ __forceinline__ void seam_tables_wave(uint8_t *arena, const MeshLayout &L, MeshDesc *D) {
  typedef Rec<CP> R;
  typedef typename R::Raw Raw;
  const uint32_t lane = lane_id();
  const uint32_t F = uni(D->num_faces), NVALL = uni(D->num_all_vertices), nad = uni((uint32_t)D->num_att_data);
  const uint32_t cmask = uni((uint32_t)D->corner_mask), allmask = (1u << nad) - 1u;
  const bool val = uni((uint32_t)D->traversal_type) == 2u;
  if (nad == 0 || nad > 7 || F != L.cap_faces || NVALL > L.cap_vertices) SM_FAIL(ST_INVALID, 680);
  const SeamLayout g = seam_layout(L.cap_faces, L.cap_vertices, nad, CP);
  uint8_t *S = arena + L.seam;
  const uint32_t *frec = (const uint32_t *)(arena + L.frec);
  const uint2 *vrec = (const uint2 *)(arena + L.vrec);
  const uint8_t *pos_vflag = arena + L.vvis;                 // bit1: the vertex lies on the boundary (k_connectivity)
  uint32_t *eseam32 = (uint32_t *)(S + g.eseam);
  const uint8_t *eseam8 = S + g.eseam;                       // byte of corner c (quad coded) = eseam8[c]
  uint8_t *vseam = S + g.vseam;
  uint32_t *pbase = (uint32_t *)(S + g.pbase);
  int32_t *c2p = (int32_t *)(arena + L.faces);
  uint8_t *blk0 = seam_block(arena, L, g, 0);
  uint32_t *list = (uint32_t *)(blk0 + g.d2c);               // seam vertices, dense (the traversal's entry -> corner: written later)
#define SM_BLK(d_) (blk0 + (uint64_t)(d_) * g.data_stride)
#define SM_VBASE(d_) ((uint32_t *)(SM_BLK(d_) + g.vbase))
#define SM_REC(d_) ((uint32_t *)(SM_BLK(d_) + g.rec))
  auto corner_ok = [&](uint32_t c) -> bool { return c < 4u * F && (c & 3u) != 3u; };

  // ---- A. faces: which seam bit belongs to which corner (one bit per interior edge, at the lower of its two faces, in corner
  // order: a prefix count), and from the bits the seam mask of the edge -- written to the bytes of BOTH corners the edge lies
  // opposite of (every byte has one writer: the edge's owner, or the corner itself on the boundary; k_init zeroed the words) --;
  // the vertices at the ends of cut edges are marked.  (Every pass of this kernel keeps four chunks of 64 elements in flight: a
  // lone wave waits a microsecond for each dependent load, and there are ten passes.)
  {
    uint8_t *eseam_w = S + g.eseam;
    uint32_t base = 0;
    bool bad = false, weird = false;
    for (uint32_t f0 = 0; f0 < F; f0 += SM_U * WAVE) {
      Raw r[SM_U];
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u) { const uint32_t f = f0 + u * WAVE + lane; r[u] = R::load(frec, f < F ? f : 0u); }
      uint32_t o[SM_U][3], idx[SM_U];
      bool own[SM_U][3];
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u) {
        const uint32_t f = f0 + u * WAVE + lane;
        const bool live = f < F;
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t k = 0; k < 3; ++k) {
          o[u][k] = live ? R::opp(r[u], k) : DSA_INVALID;
          const bool has = o[u][k] != DSA_INVALID;
          if (has && !corner_ok(o[u][k])) { bad = true; o[u][k] = DSA_INVALID; }
          if (has && (o[u][k] >> 2) == f) weird = true;      // a face glued to itself: the general path's business
          own[u][k] = o[u][k] != DSA_INVALID && (o[u][k] >> 2) >= f;
          cnt += own[u][k] ? 1u : 0u;
        }
        uint32_t total;
        idx[u] = base + wave_excl_scan(cnt, &total);
        base += total;
      }
      // the bits of the owned corners: every load of the four chunks issued before the first is used
      uint32_t bitsv[SM_U][3];
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u) {
        uint32_t ix = idx[u];
#pragma unroll
        for (uint32_t k = 0; k < 3; ++k) {
          uint32_t m = 0;
          if (own[u][k]) {
            for (uint32_t d = 0; d < nad; ++d) m |= ((((const uint32_t *)(SM_BLK(d) + g.bits))[ix >> 5] >> (ix & 31u)) & 1u) << d;
            ++ix;
          }
          bitsv[u][k] = m;
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u) {
        const uint32_t f = f0 + u * WAVE + lane;
        if (f >= F) continue;
#pragma unroll
        for (uint32_t k = 0; k < 3; ++k) {
          const bool boundary = o[u][k] == DSA_INVALID;      // boundary edges are seams of every attribute data (:516-522)
          if (!boundary && !own[u][k]) continue;             // (the owner writes this corner's byte)
          const uint32_t m = boundary ? 0xFFu : bitsv[u][k];
          if (m) {
            eseam_w[4u * f + k] = (uint8_t)m;
            if (!boundary) eseam_w[o[u][k]] = (uint8_t)m;
            // the edge opposite corner k: its end points are at the other two corners
            const uint32_t va = R::vertex(r[u], k_next(k)), vb = R::vertex(r[u], k_prev(k));
            if (va < NVALL) vseam[va] = 1;
            if (vb < NVALL) vseam[vb] = 1;
          }
        }
      }
    }
    if (__ballot(bad)) SM_FAIL(ST_INVALID, 681);
    if (__ballot(weird)) SM_FAIL(ST_NOTIMPL, DSA_SITE_RETRY_GENERAL);
    if (2u * base != D->interior_corners) SM_FAIL(ST_INVALID, 263);      // (the census of k_faces / k_seal, seen from here)
  }
  SM_SYNC();
  // ---- P0. vertices in order: one attribute vertex and one point each, unless a seam or the boundary touches the vertex (those go
  // on a list) or it has no corner at all (none)
  uint32_t nlist = 0;
  {
    bool bad = false, full = false;          // full: wave-uniform
    for (uint32_t v0 = 0; v0 < NVALL && !full; v0 += SM_U * WAVE) {
      uint32_t x[SM_U], sv[SM_U];
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u) { const uint32_t v = v0 + u * WAVE + lane, vc = v < NVALL ? v : 0u; x[u] = v < NVALL ? vrec[vc].x : DSA_INVALID; sv[u] = vseam[vc]; }
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u) {
        const uint32_t v = v0 + u * WAVE + lane;
        const bool live = v < NVALL;
        const uint32_t lm = x[u] == DSA_INVALID ? DSA_INVALID : (val ? (x[u] & 0x1FFFFFu) : x[u]);
        if (lm != DSA_INVALID && !corner_ok(lm)) bad = true;
        const bool has = lm != DSA_INVALID;
        const bool seamv = has && sv[u] != 0;
        const uint64_t m = __ballot(seamv);
        if (nlist + (uint32_t)__popcll(m) > 3u * F) { full = true; break; }      // (more vertices with a corner than corners)
        if (seamv) list[nlist + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = v;
        nlist += (uint32_t)__popcll(m);
        if (live && !seamv) {
          const uint32_t one = has ? 1u : 0u;
          for (uint32_t d = 0; d < nad; ++d) if ((cmask >> d) & 1u) SM_VBASE(d)[v] = one;
          pbase[v] = one;
        }
      }
    }
    if (__ballot(bad) || full) SM_FAIL(ST_INVALID, 651);
  }
  SM_SYNC();
  // ---- P1. the listed vertices: a walk around each counts the cuts per attribute data (k_d cuts make max(k_d, 1) attribute
  // vertices), then the cuts that separate points
  {
    bool bad = false, long_ring = false;
    for (uint32_t i0 = 0; i0 < nlist; i0 += WAVE) {
      const uint32_t i = i0 + lane;
      if (i >= nlist) continue;
      const uint32_t v = list[i];
      const uint32_t x = vrec[v].x, lm = val ? (x & 0x1FFFFFu) : x;
      const bool open = (pos_vflag[v] & 2u) != 0;
      uint64_t cnt_lo = 0, cnt_hi = 0;                       // eight 16-bit counters
      uint32_t c = lm, len = 0;
      for (;;) {
        const uint32_t xm = (uint32_t)eseam8[qnext(c)] & allmask;
        cnt_lo += (uint64_t)(xm & 1u) | ((uint64_t)((xm >> 1) & 1u) << 16) | ((uint64_t)((xm >> 2) & 1u) << 32) | ((uint64_t)((xm >> 3) & 1u) << 48);
        cnt_hi += (uint64_t)((xm >> 4) & 1u) | ((uint64_t)((xm >> 5) & 1u) << 16) | ((uint64_t)((xm >> 6) & 1u) << 32) | ((uint64_t)((xm >> 7) & 1u) << 48);
        if (++len > 60000u) { long_ring = true; break; }     // (the counters are 16 bits wide: such a fan goes to the general path)
        const uint32_t nx = pos_swing_right<CP>(frec, c);
        if (nx == DSA_INVALID || nx == lm) break;
        if (!corner_ok(nx)) { bad = true; break; }
        c = nx;
      }
      if (bad || long_ring) continue;
      uint32_t kmask = 0, E = open ? allmask : 0u;
      for (uint32_t d = 0; d < nad; ++d) {
        const uint32_t k = (uint32_t)(((d < 4 ? cnt_lo : cnt_hi) >> (16u * (d & 3u))) & 0xFFFFu);
        if ((cmask >> d) & 1u) SM_VBASE(d)[v] = k > 1u ? k : 1u;
        if (k >= 1u) kmask |= 1u << d;
        if (k >= 2u) E |= 1u << d;
      }
      // points: cuts of the data in E -- on the boundary every cut separates two attribute vertices, inside only a datum with two
      // or more cuts has different vertices on the two sides of a cut
      uint32_t kE = 0;
      if (E) {
        c = lm;
        for (uint32_t step = 0; step < len; ++step) {
          if ((uint32_t)eseam8[qnext(c)] & E) ++kE;
          const uint32_t nx = pos_swing_right<CP>(frec, c);
          if (nx == DSA_INVALID || nx == lm) break;
          c = nx;
        }
      }
      pbase[v] = kE > 1u ? kE : 1u;
      vseam[v] = (uint8_t)(kmask | 0x80u);                   // non-zero: "listed"; bit d (< 7): attribute data d has a cut here
    }
    if (__ballot(bad)) SM_FAIL(ST_INVALID, 651);
    if (__ballot(long_ring)) SM_FAIL(ST_NOTIMPL, DSA_SITE_RETRY_GENERAL);
  }
  SM_SYNC();
  // ---- P2. counts -> first ids (exclusive prefix sums over the vertices, one per attribute data and one for the points); the
  // boundary flag of every attribute vertex (MeshAttributeCornerTable.cs IsOnBoundary: a cut or the mesh boundary at the vertex)
  uint32_t totals[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ptotal = 0;
  for (uint32_t v0 = 0; v0 < NVALL; v0 += SM_U * WAVE) {
    uint32_t km[SM_U], np[SM_U];
#pragma unroll
    for (uint32_t u = 0; u < SM_U; ++u) { const uint32_t v = v0 + u * WAVE + lane; km[u] = v < NVALL ? (uint32_t)vseam[v] : 0u; np[u] = v < NVALL ? pbase[v] : 0u; }
    for (uint32_t d = 0; d < nad; ++d) {
      if (!((cmask >> d) & 1u)) continue;
      uint32_t *vb = SM_VBASE(d);
      uint8_t *avf = SM_BLK(d) + g.vflag;
      uint32_t nn[SM_U];
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u) { const uint32_t v = v0 + u * WAVE + lane; nn[u] = v < NVALL ? vb[v] : 0u; }
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u) {
        const uint32_t v = v0 + u * WAVE + lane;
        uint32_t total;
        const uint32_t first = totals[d] + wave_excl_scan(nn[u], &total);
        totals[d] += total;
        if (v < NVALL) vb[v] = first;
        if (first + nn[u] <= 3u * F) {
          const uint8_t fl = ((km[u] >> d) & 1u) ? 2 : 0;
          for (uint32_t j = 0; j < nn[u]; ++j) avf[first + j] = fl;
        }
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < SM_U; ++u) {
      const uint32_t v = v0 + u * WAVE + lane;
      uint32_t total;
      const uint32_t first = ptotal + wave_excl_scan(np[u], &total);
      ptotal += total;
      if (v < NVALL) pbase[v] = first;
    }
  }
  for (uint32_t d = 0; d < nad; ++d) if (((cmask >> d) & 1u) && totals[d] > 3u * F) SM_FAIL(ST_INVALID, 650);
  if (ptotal > L.cap_points) SM_FAIL(ST_INVALID, 662);
  SM_SYNC();
  // ---- P4. faces: the virtual records of the corner-attribute data -- opposites cut at the attribute's seams, vertices: the one
  // id of every vertex that is not listed (the listed ones' fields stay "unset", all ones, for the walks of P3) -- and the one
  // point of those vertices
  {
    bool bad = false;
    for (uint32_t f0 = 0; f0 < F; f0 += SM_U * WAVE) {
      Raw r[SM_U];
      uint32_t word[SM_U];
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u) { const uint32_t f = f0 + u * WAVE + lane, fc = f < F ? f : 0u; r[u] = R::load(frec, fc); word[u] = eseam32[fc]; }
      uint32_t sv[SM_U][3];
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u) {
        const uint32_t f = f0 + u * WAVE + lane;
#pragma unroll
        for (uint32_t k = 0; k < 3; ++k) {
          const uint32_t v = R::vertex(r[u], k);
          if (f < F && v >= NVALL) bad = true;
          sv[u][k] = (f < F && v < NVALL) ? (uint32_t)vseam[v] : 1u;
        }
      }
      uint32_t pb[SM_U][3];
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u)
#pragma unroll
        for (uint32_t k = 0; k < 3; ++k) pb[u][k] = sv[u][k] == 0 ? pbase[R::vertex(r[u], k)] : 0u;
      for (uint32_t d = 0; d < nad; ++d) {
        if (!((cmask >> d) & 1u)) continue;
        const uint32_t *vb = SM_VBASE(d);
        uint32_t *rec = SM_REC(d);
        uint32_t id[SM_U][3];
#pragma unroll
        for (uint32_t u = 0; u < SM_U; ++u)
#pragma unroll
          for (uint32_t k = 0; k < 3; ++k) id[u][k] = sv[u][k] == 0 ? vb[R::vertex(r[u], k)] : DSA_INVALID;
#pragma unroll
        for (uint32_t u = 0; u < SM_U; ++u) {
          const uint32_t f = f0 + u * WAVE + lane;
          if (f >= F) continue;
          uint32_t oc[3];
#pragma unroll
          for (uint32_t k = 0; k < 3; ++k) oc[k] = ((word[u] >> (8u * k + d)) & 1u) ? DSA_INVALID : R::opp(r[u], k);
          if (CP) {
            const uint64_t vv = Rec<true>::pack(id[u][0], id[u][1], id[u][2]) | (~0ull << 63), oo = Rec<true>::pack(oc[0], oc[1], oc[2]) | (~0ull << 63);
            ((uint4 *)rec)[f] = make_uint4((uint32_t)vv, (uint32_t)(vv >> 32), (uint32_t)oo, (uint32_t)(oo >> 32));
          } else {
            ((uint4 *)rec)[(size_t)f * 2] = make_uint4(id[u][0], id[u][1], id[u][2], 0u);
            ((uint4 *)rec)[(size_t)f * 2 + 1] = make_uint4(oc[0], oc[1], oc[2], 0u);
          }
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < SM_U; ++u) {
        const uint32_t f = f0 + u * WAVE + lane;
        if (f >= F) continue;
#pragma unroll
        for (uint32_t k = 0; k < 3; ++k) if (sv[u][k] == 0) c2p[3u * f + k] = (int32_t)pb[u][k];
      }
    }
    if (__ballot(bad)) SM_FAIL(ST_INVALID, 681);
  }
  SM_SYNC();
  // ---- P3. the listed vertices again: ids per corner by a walk.  With cum = cuts met so far (the corner's own left edge
  // included) and s = 1 if the walk starts on a cut: attribute vertex = first + (cum - s) mod n (RecomputeVertices numbers from the
  // corner behind the last cut, :116-152); point = first + (cum_E - s_E) mod n_P, where the count starts at the corner
  // AssignPointsToCorners starts from (:559-590: the first corner whose attribute vertex differs from the left-most corner's, for
  // the first datum that has such a corner; the left-most corner itself on the boundary)
  {
    bool bad = false;
    for (uint32_t i0 = 0; i0 < nlist; i0 += WAVE) {
      const uint32_t i = i0 + lane;
      if (i >= nlist) continue;
      const uint32_t v = list[i];
      const uint32_t x = vrec[v].x, lm = val ? (x & 0x1FFFFFu) : x;
      const bool open = (pos_vflag[v] & 2u) != 0;
      uint32_t first[8], n[8];
      uint32_t E = open ? allmask : 0u, dstar = 0xFFu;
      for (uint32_t d = 0; d < 8; ++d) { first[d] = 0; n[d] = 1; }
      for (uint32_t d = 0; d < nad; ++d) if ((cmask >> d) & 1u) first[d] = SM_VBASE(d)[v];
      // counts again from the walk (cheaper than a second array): k_d, and from them n_d, E, d*
      uint64_t cnt_lo = 0, cnt_hi = 0;
      uint32_t c = lm, len = 0;
      for (;;) {
        const uint32_t xm = (uint32_t)eseam8[qnext(c)] & allmask;
        cnt_lo += (uint64_t)(xm & 1u) | ((uint64_t)((xm >> 1) & 1u) << 16) | ((uint64_t)((xm >> 2) & 1u) << 32) | ((uint64_t)((xm >> 3) & 1u) << 48);
        cnt_hi += (uint64_t)((xm >> 4) & 1u) | ((uint64_t)((xm >> 5) & 1u) << 16) | ((uint64_t)((xm >> 6) & 1u) << 32) | ((uint64_t)((xm >> 7) & 1u) << 48);
        ++len;
        const uint32_t nx = pos_swing_right<CP>(frec, c);
        if (nx == DSA_INVALID || nx == lm || len > 60000u) break;
        c = nx;
      }
      for (uint32_t d = 0; d < nad; ++d) {
        const uint32_t k = (uint32_t)(((d < 4 ? cnt_lo : cnt_hi) >> (16u * (d & 3u))) & 0xFFFFu);
        n[d] = k > 1u ? k : 1u;
        if (k >= 2u) { E |= 1u << d; if (!open && dstar == 0xFFu) dstar = d; }
      }
      // where the points start: s_E = cuts of E up to and including the start corner
      uint32_t sE = 1, nP = 1;
      if (E) {
        uint32_t kE = 0, cumE = 0;
        bool found = open;                                   // on the boundary the walk starts at the left-most corner, a cut of E itself
        c = lm;
        for (uint32_t step = 0; step < len; ++step) {
          const uint32_t xm = (uint32_t)eseam8[qnext(c)];
          if (xm & E) { ++kE; }
          if (!found && step >= 1u && ((xm >> dstar) & 1u)) { found = true; cumE = kE; }
          const uint32_t nx = pos_swing_right<CP>(frec, c);
          if (nx == DSA_INVALID || nx == lm) break;
          c = nx;
        }
        nP = kE > 1u ? kE : 1u;
        sE = open ? 1u : (found ? cumE : 0u);
      }
      // the assignment
      const uint32_t pfirst = pbase[v];
      uint32_t cum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s[8], cumE = 0;
      c = lm;
      for (uint32_t step = 0; step < len; ++step) {
        const uint32_t xm = (uint32_t)eseam8[qnext(c)];
        if (step == 0) for (uint32_t d = 0; d < 8; ++d) s[d] = (xm >> d) & 1u;
        for (uint32_t d = 0; d < nad; ++d) {
          cum[d] += (xm >> d) & 1u;
          if (!((cmask >> d) & 1u)) continue;
          uint32_t off = cum[d] - s[d];
          off = off >= n[d] ? off - n[d] : off;
          const uint32_t id = first[d] + off;
          if (CP) vrec_set_compact(SM_REC(d), c, id);
          else SM_REC(d)[fv_idx(c)] = id;
        }
        if (xm & E) ++cumE;
        uint32_t poff = E ? cumE + nP - sE : 0u;
        while (poff >= nP) poff -= nP;
        c2p[3u * (c >> 2) + (c & 3u)] = (int32_t)(pfirst + poff);
        const uint32_t nx = pos_swing_right<CP>(frec, c);
        if (nx == DSA_INVALID || nx == lm) break;
        c = nx;
      }
    }
    if (__ballot(bad)) SM_FAIL(ST_INVALID, 651);
  }
  // ---- results: points of the mesh, vertices of every attribute table = entries of its decoder's attributes
  if (lane == 0) {
    D->num_points = ptotal;
    D->seam_tables_done = 1;
    for (uint32_t d = 0; d < nad; ++d) D->seam_nv[d] = totals[d];
    for (uint32_t ai = 0; ai < D->num_attributes; ++ai) {
      AttrDesc &a = D->att[ai];
      if (a.corner_data == 0) continue;
      const uint32_t e = totals[(uint32_t)a.corner_data - 1u];
      a.num_entries = e;
      const uint64_t nvals = (uint64_t)e * a.nc_portable, bytes = (uint64_t)e * a.nc * data_type_length(a.data_type);
      if (nvals > L.work_cap[ai] || bytes > L.out_cap[ai]) { fail(D, ST_INVALID, 142); break; }
    }
  }
#undef SM_BLK
#undef SM_VBASE
#undef SM_REC
}

__global__ __launch_bounds__(WAVE) void k_seam_tables(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  const uint32_t mesh = blockIdx.x;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (status_of(D) != ST_OK || D->general || !D->seam_fast || D->encoder_type == 0) return;
  const MeshLayout &L = layouts[mesh];
  if (L.seam_bytes == 0) return;
  __builtin_amdgcn_s_setprio(DSA_CHAIN_PRIO);              // on the critical path of a seamed mesh, like the connectivity before it
  if (L.rec_compact) seam_tables_wave<true>(arena, L, D);
  else seam_tables_wave<false>(arena, L, D);
}

// k_traverse_att: the depth-first order of every corner-attribute decoder on its own table -- one wave per (mesh, attribute
// data), the program of k_traverse on the virtual mesh -- and the parallelogram operands of its entries.
__global__ __launch_bounds__(WAVE) void k_traverse_att(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  __shared__ unsigned long long sh_tf[TR_SLOTS], sh_tv[TR_SLOTS];
  __shared__ uint32_t sh_hist[TR_HIST_WORDS];
  for (uint32_t i = threadIdx.x; i < TR_SLOTS; i += WAVE) { sh_tf[i] = 0; sh_tv[i] = 0; }
  if (threadIdx.x < TR_HIST_WORDS) sh_hist[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t mesh = blockIdx.x, d = blockIdx.y;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (status_of(D) != ST_OK || D->general || !D->seam_fast || d >= D->num_att_data || !((D->corner_mask >> d) & 1u)) return;
  const MeshLayout &L = layouts[mesh];
  const TravIO io = trav_attribute(arena, L, D, d);
  if (L.rec_compact) traverse_wave<true>(arena, L, D, io, 1u | 2u, sh_tf, sh_tv, sh_hist);
  else traverse_wave<false>(arena, L, D, io, 1u | 2u, sh_tf, sh_tv, sh_hist);
}

// k_seam_maps: point -> entry of every attribute of a mesh with corner attributes (MeshTraversalSequencer.cs:33-50), from the
// corners: the point of a corner, the attribute's vertex at that corner, the entry that vertex was given.  All corners of a point
// carry the same vertex of every table (that is what made them one point), so concurrent writers agree.
__global__ __launch_bounds__(256) void k_seam_maps(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  const uint32_t mesh = blockIdx.y;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (status_of(D) != ST_OK || D->general || !D->seam_fast) return;
  const MeshLayout &L = layouts[mesh];
  const uint32_t F = D->num_faces, na = D->num_attributes, npts = D->num_points, nad = D->num_att_data;
  const SeamLayout g = seam_layout(L.cap_faces, L.cap_vertices, nad, L.rec_compact != 0);
  const uint32_t *frec = (const uint32_t *)(arena + L.frec);
  const int32_t *c2p = (const int32_t *)(arena + L.faces);
  const int32_t *v2d_pos = (const int32_t *)(arena + L.v2d);
  const bool compact = L.rec_compact != 0;
  bool bad = false;
  for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < F; f += gridDim.x * blockDim.x) {
    const uint4 pv = compact ? Rec<true>::vertices_of(frec, f) : Rec<false>::vertices_of(frec, f);
    const uint32_t pvert[3] = {pv.x, pv.y, pv.z};
    uint32_t point[3];
    for (uint32_t k = 0; k < 3; ++k) { point[k] = (uint32_t)c2p[3 * f + k]; if (point[k] >= npts) bad = true; }
    for (uint32_t ai = 0; ai < na; ++ai) {
      const AttrDesc &a = D->att[ai];
      uint32_t vert[3];
      const int32_t *v2d = v2d_pos;
      uint32_t nverts = D->num_vertices;
      if (a.corner_data) {
        const uint32_t d = (uint32_t)a.corner_data - 1u;
        const uint8_t *blk = seam_block(arena, L, g, d);
        const uint4 av = compact ? Rec<true>::vertices_of((const uint32_t *)(blk + g.rec), f) : Rec<false>::vertices_of((const uint32_t *)(blk + g.rec), f);
        vert[0] = av.x; vert[1] = av.y; vert[2] = av.z;
        v2d = (const int32_t *)(blk + g.v2d);
        nverts = D->seam_nv[d];
      } else { vert[0] = pvert[0]; vert[1] = pvert[1]; vert[2] = pvert[2]; }
      uint32_t *map = (uint32_t *)(arena + L.map[ai]);
      for (uint32_t k = 0; k < 3; ++k) {
        if (vert[k] >= nverts || point[k] >= npts) { bad = true; continue; }
        const int32_t e = v2d[vert[k]];
        if (e < 0 || (uint32_t)e >= npts) { bad = true; continue; }
        map[point[k]] = (uint32_t)e;
      }
    }
  }
  if (__ballot(bad) && lane_id() == 0) fail(D, ST_INVALID, 670);
}

#undef SM_FAIL
#undef SM_SYNC
#undef SM_U

// =========================================================================
// TexCoordsPortable (MeshPredictionSchemeTexCoordsPortableDecoder.cs:50-85, ...PortablePredictor.cs:46-150; what stock encoders
// pick for texture coordinates at their default level) on the fast kernels, for meshes with and without seams.
// The prediction of an entry reads the DECODED texture coordinates of the entries at Next / Previous of its corner: a serial chain
// over the entries of an attribute.  But everything else it needs -- which entries those are, the positions of the three
// corners, the squared edge length, the foot of the tip on the edge, the 64-bit integer square root -- depends on the mesh and
// on the positions only.  So:
//   k_flip_bits (extended)   the orientation bits (one serial rABS stream per attribute), from the start of the decode
//   k_texcoords_prepare      one thread per entry, behind the traversals and the prediction of the positions: TcPrep
//   k_texcoords              one LANE per attribute (64 attributes to a wave: no cross-lane traffic in the chain, so the
//                            instruction stream is shared): per entry two multiply-adds, two divisions by an estimate put
//                            right by remainders, the wrap transform
// =========================================================================
__device__ __forceinline__ uint64_t isqrt_floor(uint64_t n) {   // = Core/MathUtilities.cs:5-25 IntSqrt (its Newton iteration ends on floor(sqrt(n)))
  if (n == 0) return 0;
  uint64_t r = (uint64_t)__dsqrt_rn((double)n);
  if (r > 0xFFFFFFFFull) r = 0xFFFFFFFFull;
  while (r * r > n) --r;
  while (r < 0xFFFFFFFFull && (r + 1) * (r + 1) <= n) ++r;
  return r;
}
__device__ __forceinline__ int64_t div_trunc(int64_t x, int64_t y) { return y > 0 ? div_trunc_pos(x, y) : x / y; }

// The orientation bits as the decoder will use them (:66-85: a bit says "same as the one before", starting from true).
__global__ __launch_bounds__(WAVE) void k_orient_bits(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t lanes_per_mesh, uint32_t late_pass) {
  const uint32_t lane = lane_id();
  const uint32_t mesh = blockIdx.x * (WAVE / lanes_per_mesh) + lane / lanes_per_mesh, ai = lane % lanes_per_mesh;
  if (mesh >= n) return;
  const MeshLayout &L = layouts[mesh];
  MeshDesc *D = &descs[mesh];
  if (status_of(D) != ST_OK || D->general || ai >= D->num_attributes) return;
  const AttrDesc &a = D->att[ai];
  // late_pass: 1 the attributes the walk located only behind k_seam_tables (a corner attribute's own orientation block is known from
  // the start: its place does not depend on the entry count once the walk got there)
  if (!a.have_scheme || a.pred_kind != 3 || a.source == SRC_BYTES || (a.late_located != 0) != (late_pass != 0)) return;
  if (L.tc[ai] == 0) { fail(D, ST_NOTIMPL, DSA_SITE_RETRY_GENERAL); return; }
  Rabs rb;
  uint32_t endp;
  rb.start(arena + L.stream, L.stream_len, a.off_flips, &endp);
  if (!rb.ok) { fail(D, ST_INVALID, 669); return; }
  uint32_t cap;
  uint32_t *bits = orient_bits_of(arena, L, D, ai, &cap);
  const uint32_t count = a.num_orient;
  if (count > cap) { fail(D, ST_INVALID, 668); return; }
  (void)rabs_block_to_words<true>(rb, count, bits);
}

template <bool CP>
__device__ __forceinline__ void texcoords_prepare_entries(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t pa, uint32_t tid, uint32_t stride) {
  typedef Rec<CP> R;
  const AttrDesc &a = D->att[ai];
  const uint32_t entries = a.num_entries, F = D->num_faces;
  TravIO io = a.corner_data ? trav_attribute(arena, L, D, (uint32_t)a.corner_data - 1u) : trav_position(arena, L, D);
  const uint32_t *frec_pos = (const uint32_t *)(arena + L.frec);
  const int32_t *v2d_pos = (const int32_t *)(arena + L.v2d);
  const int32_t *posv = (const int32_t *)(arena + L.work[pa]);
  const uint32_t pos_entries = D->att[pa].num_entries, NVP = D->num_vertices;
  TcPrep *prep = (TcPrep *)(arena + L.tc[ai]);
  bool bad = false;
  for (uint32_t p = tid; p < entries; p += stride) {
    TcPrep t;
    t.next_id = DSA_INVALID; t.prev_id = DSA_INVALID; t.pn_norm2 = 0; t.cn_dot_pn = 0; t.norm = 0; t.inv = 0.0;
    const uint32_t ci = io.d2c[p];
    if ((ci >> 2) >= F || (ci & 3u) == 3u) { bad = true; prep[p] = t; continue; }
    const uint32_t cn = qnext(ci), cp = qprev(ci);
    const uint32_t vn = R::get_v(io.frec, cn), vp = R::get_v(io.frec, cp);
    if (vn >= io.NV || vp >= io.NV) { bad = true; prep[p] = t; continue; }
    const int32_t en = io.v2d[vn], ep = io.v2d[vp];
    if (en >= 0 && (uint32_t)en < p) t.next_id = (uint32_t)en;
    if (ep >= 0 && (uint32_t)ep < p) t.prev_id = (uint32_t)ep;
    if (t.next_id != DSA_INVALID && t.prev_id != DSA_INVALID) {
      // the position of an entry of THIS attribute = the position at the corner the entry was reached through
      auto position = [&](uint32_t corner, int64_t dst[3]) {
        const uint32_t v = (corner >> 2) < F && (corner & 3u) != 3u ? R::get_v(frec_pos, corner) : DSA_INVALID;
        const int32_t e = v < NVP ? v2d_pos[v] : -1;
        if (e < 0 || (uint32_t)e >= pos_entries) { bad = true; dst[0] = dst[1] = dst[2] = 0; return; }
        for (int k = 0; k < 3; ++k) dst[k] = posv[(size_t)e * 3 + k];
      };
      int64_t tip[3], np[3], pp[3];
      position(ci, tip); position(io.d2c[t.next_id], np); position(io.d2c[t.prev_id], pp);
      // (64-bit wrap-around arithmetic, as the reference's)
      uint64_t pn[3], cnv[3];
      for (int k = 0; k < 3; ++k) { pn[k] = (uint64_t)pp[k] - (uint64_t)np[k]; cnv[k] = (uint64_t)tip[k] - (uint64_t)np[k]; }
      const int64_t pn_norm2 = (int64_t)(pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2]);
      t.pn_norm2 = pn_norm2;
      t.inv = __drcp_rn((double)(uint32_t)pn_norm2);         // (taken by the chain only where pn_norm2 fits 32 bits and is not zero)
      if (pn_norm2 != 0) {
        const int64_t cn_dot_pn = (int64_t)(pn[0] * cnv[0] + pn[1] * cnv[1] + pn[2] * cnv[2]);
        t.cn_dot_pn = cn_dot_pn;
        uint64_t cx2 = 0;
        for (int k = 0; k < 3; ++k) {
          const int64_t x_pos = (int64_t)((uint64_t)np[k] + (uint64_t)div_trunc((int64_t)((uint64_t)cn_dot_pn * pn[k]), pn_norm2));
          const uint64_t cx = (uint64_t)tip[k] - (uint64_t)x_pos;
          cx2 += cx * cx;
        }
        t.norm = (int64_t)isqrt_floor(cx2 * (uint64_t)pn_norm2);
      }
    }
    prep[p] = t;
  }
  if (bad) fail(D, ST_INVALID, 671);
}
__global__ __launch_bounds__(256) void k_texcoords_prepare(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  const uint32_t mesh = blockIdx.y, ai = blockIdx.z;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (status_of(D) != ST_OK || D->general || ai >= D->num_attributes) return;
  const AttrDesc &a = D->att[ai];
  const MeshLayout &L = layouts[mesh];
  if (!a.have_scheme || a.pred_kind != 3 || a.source == SRC_BYTES || a.num_entries == 0 || L.tc[ai] == 0) return;
  int pa = -1;                                              // parent: the portable positions (k_locate checked that they are there)
  for (uint32_t k = 0; k < ai; ++k) if (D->att[k].att_type == 0 && D->att[k].seq_type != 0) { pa = (int)k; break; }
  if (pa < 0) { fail(D, ST_INVALID, 667); return; }
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  if (L.rec_compact) texcoords_prepare_entries<true>(arena, L, D, ai, (uint32_t)pa, tid, stride);
  else texcoords_prepare_entries<false>(arena, L, D, ai, (uint32_t)pa, tid, stride);
}

// The chain: one lane per (mesh, attribute).  A step needs the entry's TcPrep record, its correction and the decoded texture
// coordinates of two earlier entries -- loads that a lone lane would wait a microsecond for, 33 000 times.  None of their addresses
// depends on the chain, so they are requested ahead: entries are taken in groups of TC_G; while group g is computed the records of
// group g + 2 and the operands of group g + 1 are on their way.  An operand that is not final when it would have to be requested
// (an entry of group g or g + 1 itself: the usual case for one of the two, the strip's previous entry) comes from a register window
// of the last 2 TC_G results instead.
#define TC_G 4
// su / d truncated toward zero (the C# operator) for the chain: the quotient from a double multiplication with 1 / d, put right by
// the remainder -- exact while d fits 32 bits and the quotient 31 (anything else takes the operator's long road)
__device__ __noinline__ int64_t tc_div_slow(int64_t x, int64_t y) { return x / y; }      // (one copy: the chain is unrolled twelve entries deep)
__device__ __forceinline__ int32_t tc_div(int64_t su, int64_t d, double inv, bool d_small) {
  const double sd = __fma_rn((double)(int32_t)(su >> 32), 4294967296.0, (double)(uint32_t)su);      // exact below 2^53
  const double qe = sd * inv;
  if (!d_small || !(qe > -2147483000.0 && qe < 2147483000.0) || !(sd > -4.0e15 && sd < 4.0e15)) return (int32_t)tc_div_slow(su, d);
  // toward zero; off by one at most: sd is exact, `inv` and the product carry one rounding each (2^-53 relative), the quotient is
  // below 2^31 -- the estimate is within 2^-20 of the true quotient, so its truncation is the true one or its neighbour
  int32_t q = (int32_t)qe;
  const int64_t r = su - (int64_t)q * d;
  if (su >= 0) q += r < 0 ? -1 : (r >= d ? 1 : 0);
  else q += r > 0 ? 1 : (r <= -d ? -1 : 0);
  return q;
}
// The loop body has no memory load inside a branch: every request is issued in straight-line code, so that the waits the compiler
// places count exactly the younger requests (vmcnt is one in-order counter) instead of draining everything that is in flight.
#define TC_DEPTH 3        // groups of records in flight
#define TC_RING 8         // results of a lane kept in LDS (the operands too recent to have been requested: < TC_G + TC_G back)
// TWO lanes per attribute, one for u and one for v (lanes 2j and 2j + 1): the two halves of the prediction are the same program on
// different components, and what one needs of the other -- the other component of pn_uv, whether its coordinates were equal too --
// is a neighbour exchange in the quad (one DPP move).  Half the instructions per entry on the chain.
__device__ __forceinline__ uint32_t tc_partner(uint32_t x) { return dpp_mov<0xB1>(x); }      // quad_perm [1, 0, 3, 2]
__global__ __launch_bounds__(WAVE) void k_texcoords(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  __shared__ int32_t ring[TC_RING][WAVE];
  __builtin_amdgcn_s_setprio(3);              // a handful of long chains: issue ahead of whatever shares the SIMD
  const uint32_t lane = lane_id();
  const uint32_t mesh = (blockIdx.x * WAVE + lane) >> 1, comp = lane & 1u, ai = blockIdx.y;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (status_of(D) != ST_OK || D->general || ai >= D->num_attributes) return;
  const AttrDesc &a = D->att[ai];
  const MeshLayout &L = layouts[mesh];
  if (!a.have_scheme || a.pred_kind != 3 || a.source == SRC_BYTES || a.num_entries == 0 || L.tc[ai] == 0) return;
  const TcPrep *prep = (const TcPrep *)(arena + L.tc[ai]);
  int32_t *w = (int32_t *)(arena + L.work[ai]) + comp;      // this lane's component of entry e: w[2 e]
  uint32_t cap;
  const uint32_t *obits = orient_bits_of(arena, L, D, ai, &cap);
  const uint32_t entries = a.num_entries, lastp = entries - 1;
  const int32_t mn = a.wrap_min, mx = a.wrap_max, max_dif = 1 + mx - mn;
  // Orientations are taken from the back of the list (:111-113).  The word in use and the two below it are kept in registers; the
  // two lower ones are requested anew at every group's turn (for the word index of that moment) and taken over at the next turn,
  // where the index has moved by one word at most (a group consumes four orientations at most).
  uint32_t left = a.num_orient;
  uint32_t ow_idx = left ? (left - 1) >> 5 : 0u;
  uint32_t ow_cur = left ? obits[ow_idx] : 0u, ow_n1 = obits[ow_idx ? ow_idx - 1 : 0u], ow_n2 = obits[ow_idx > 1 ? ow_idx - 2 : 0u];
  uint32_t ldA = ow_n1, ldB = ow_n2, ld_base = ow_idx;
  bool ran_out = false;
  TcPrep rt[TC_DEPTH][TC_G];
  int32_t rc[TC_DEPTH][TC_G], cfn[TC_G], cfp[TC_G], nfn[TC_G], nfp[TC_G];
  int32_t o1 = 0;                                            // the result of entry p - 1 (this lane's component)
#pragma unroll
  for (int k = 0; k < TC_DEPTH; ++k)
#pragma unroll
    for (int u = 0; u < TC_G; ++u) {
      const uint32_t p = (uint32_t)(k * TC_G + u) <= lastp ? (uint32_t)(k * TC_G + u) : lastp;
      rt[k][u] = prep[p]; rc[k][u] = w[2 * (size_t)p];
    }
#pragma unroll
  for (int u = 0; u < TC_G; ++u) { nfn[u] = 0; nfp[u] = 0; }
#pragma unroll
  for (int i = 0; i < TC_RING; ++i) ring[i][lane] = 0;
  for (uint32_t gbase = 0; gbase <= lastp; gbase += TC_DEPTH * TC_G) {
#pragma unroll
    for (int k = 0; k < TC_DEPTH; ++k) {
      const uint32_t g0 = gbase + (uint32_t)k * TC_G;      // first entry of the group whose turn it is (slot k)
      const int kn = (k + 1) % TC_DEPTH;                   // (a constant once the loop is unrolled)
      // ---- the turn starts: take over what the last turn requested, request for the next one
#pragma unroll
      for (int u = 0; u < TC_G; ++u) { cfn[u] = nfn[u]; cfp[u] = nfp[u]; }
      { const bool same = ld_base == ow_idx; ow_n1 = same ? ldA : ldB; ow_n2 = same ? ldB : ow_n2; }
#pragma unroll
      for (int u = 0; u < TC_G; ++u) {                     // the next group's operands that are final by now (entries < g0)
        nfn[u] = w[2 * (size_t)(rt[kn][u].next_id < g0 ? rt[kn][u].next_id : 0u)];
        nfp[u] = w[2 * (size_t)(rt[kn][u].prev_id < g0 ? rt[kn][u].prev_id : 0u)];
      }
      ldA = obits[ow_idx ? ow_idx - 1 : 0u]; ldB = obits[ow_idx > 1 ? ow_idx - 2 : 0u]; ld_base = ow_idx;
      const uint32_t far_limit = g0 >= TC_G ? g0 - TC_G : 0u;     // what this group's request (a turn ago) covered
      // ---- the entries of the group, one after the other
#pragma unroll
      for (int u = 0; u < TC_G; ++u) {
        const uint32_t p = g0 + u;
        const bool live = p <= lastp;
        const TcPrep t = rt[k][u];
        const bool hn = t.next_id != DSA_INVALID, hp = t.prev_id != DSA_INVALID, both = hn && hp;
        // operands (this lane's component): the previous result, a recent one from the ring, or what was requested a turn ago
        const int32_t ln = ring[t.next_id & (TC_RING - 1)][lane], lp = ring[t.prev_id & (TC_RING - 1)][lane];
        const int32_t nc = p - t.next_id == 1u ? o1 : (t.next_id >= far_limit ? ln : cfn[u]);
        const int32_t pc = p - t.prev_id == 1u ? o1 : (t.prev_id >= far_limit ? lp : cfp[u]);
        const uint32_t eq_own = pc == nc ? 1u : 0u;
        const uint32_t eq_other = tc_partner(eq_own);
        const bool equal = (eq_own & eq_other) != 0;
        const int64_t d = t.pn_norm2;
        const bool geo = both && !equal && d != 0;
        const bool d_small = d > 0 && d < (int64_t)0x100000000ll;
        const double inv = t.inv;
        const int32_t d_own = (int32_t)((uint32_t)pc - (uint32_t)nc);
        const int32_t d_other = (int32_t)tc_partner((uint32_t)d_own);
        // x = n d + (cn . pn) pn_uv (own component); cx = (pn_v, -pn_u) norm: the other component's difference, negated for v
        const uint64_t x = (uint64_t)(int64_t)nc * (uint64_t)d + (uint64_t)t.cn_dot_pn * (uint64_t)(int64_t)d_own;
        const uint64_t cx = (uint64_t)(int64_t)(comp ? -(int64_t)d_other : (int64_t)d_other) * (uint64_t)t.norm;
        // the orientation (consumed by this entry only when it takes the geometric prediction); both lanes keep the same books
        const bool take = geo && live;
        ran_out = ran_out || (take && left == 0);
        const uint32_t nl = take && left ? left - 1 : left;
        const bool cross = (nl >> 5) != ow_idx && take && left;
        ow_idx = cross ? nl >> 5 : ow_idx;
        ow_cur = cross ? ow_n1 : ow_cur;
        ow_n1 = cross ? ow_n2 : ow_n1;
        left = nl;
        const bool orientation = ((ow_cur >> (left & 31u)) & 1u) != 0;
        const int64_t sx = (int64_t)(orientation ? x + cx : x - cx);
        int32_t g = 0;
        if (geo) g = tc_div(sx, d, inv, d_small);
        // the fallback chain of the predictor, as written there (:129-149): the entry at Next if it is decoded, else the entry before
        const int32_t fb = hn ? nc : (p > 0 ? o1 : 0);
        const int32_t pred = both && equal ? pc : (geo ? g : fb);
        int32_t o = wrap_original(pred, rc[k][u], mn, mx, max_dif);
        o = live ? o : o1;                                 // (behind the last entry: it is written again)
        w[2 * (size_t)(live ? p : lastp)] = o;
        ring[p & (TC_RING - 1)][lane] = o;
        o1 = o;
      }
      // ---- the slot is free: the records of the group TC_DEPTH turns ahead
#pragma unroll
      for (int u = 0; u < TC_G; ++u) {
        const uint32_t p2 = g0 + TC_DEPTH * TC_G + u <= lastp ? g0 + TC_DEPTH * TC_G + u : lastp;
        rt[k][u] = prep[p2]; rc[k][u] = w[2 * (size_t)p2];
      }
    }
  }
  if (ran_out && comp == 0) fail(D, ST_INVALID, 672);
}
#undef TC_DEPTH
#undef TC_RING
#undef TC_G


// =========================================================================
// ConstrainedMultiParallelogram (MeshPredictionSchemeConstrainedMultiParallelogramDecoder.cs:28-136; what stock encoders pick for
// positions and generic attributes at their two highest compression levels) on the fast kernels.
// The prediction of an entry is the average of up to four parallelograms -- those found while swinging left, then right, around the
// entry's vertex whose three other corners are decoded already -- each kept or dropped by a crease flag from the bit array of
// its context (= the number found - 1).  Which parallelograms an entry has, and which flags are its own, depends on the mesh and
// the traversal order only; the values form a chain (every entry reads DECODED entries).  So:
//   k_crease_bits          the four flag arrays (one serial rABS stream each, one lane per stream), from the start of the decode
//   k_multipara_prepare    one thread per entry, behind the traversal: MpPrep (the parallelograms as entry ids)
//   k_multipara            the chain, SIXTEEN lanes per attribute: lane (i, c) gathers the three operands of parallelogram i,
//                          component c from a window of the last 1024 results in LDS (the reach of a ring of a spiral or a strip of
//                          a grid: an LDS round trip per entry where memory would take a microsecond), two exchanges add the four
//                          parallelograms up, every lane divides and applies the wrap transform for its component.
// =========================================================================
__device__ __forceinline__ uint32_t *mp_crease_words_of(uint8_t *arena, const MeshLayout &L, uint32_t ai, uint32_t cap_entries) {
  return (uint32_t *)(arena + L.tc[ai] + sizeof(MpPrep) * (size_t)cap_entries);
}
__device__ __forceinline__ uint32_t mp_crease_base(const AttrDesc &a, uint32_t ctx) {       // first word of a context's flags
  uint32_t w = 0;
  for (uint32_t k = 0; k < ctx; ++k) w += (a.num_crease[k] + 31u) >> 5;
  return w;
}
__global__ __launch_bounds__(WAVE) void k_crease_bits(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, uint32_t lanes_per_mesh) {
  const uint32_t lane = lane_id();
  const uint32_t per = 4u * lanes_per_mesh;                  // (mesh, attribute, context)
  const uint32_t mesh = blockIdx.x * (WAVE / per) + lane / per, ai = (lane % per) >> 2, ctx = lane & 3u;
  if (mesh >= n) return;
  const MeshLayout &L = layouts[mesh];
  MeshDesc *D = &descs[mesh];
  if (status_of(D) != ST_OK || D->general || ai >= D->num_attributes) return;
  const AttrDesc &a = D->att[ai];
  if (!a.have_scheme || a.pred_kind != 4 || a.source == SRC_BYTES || a.num_crease[ctx] == 0) return;
  Rabs rb;
  uint32_t endp;
  rb.start(arena + L.stream, L.stream_len, a.off_crease[ctx], &endp);
  if (!rb.ok) { fail(D, ST_INVALID, 674); return; }
  (void)rabs_block_to_words<false>(rb, a.num_crease[ctx], mp_crease_words_of(arena, L, ai, L.cap_vertices) + mp_crease_base(a, ctx));
}

template <bool CP>
__device__ __forceinline__ void multipara_prepare_entries(uint8_t *arena, const MeshLayout &L, MeshDesc *D, uint32_t ai, uint32_t tid, uint32_t stride) {
  typedef Rec<CP> R;
  const AttrDesc &a = D->att[ai];
  const uint32_t entries = a.num_entries, F = D->num_faces;
  const TravIO io = trav_position(arena, L, D);
  MpPrep *prep = (MpPrep *)(arena + L.tc[ai]);
  bool bad = false;
  const uint32_t max_steps = 3u * F + 1u;
  auto opposite = [&](uint32_t c) { return c == DSA_INVALID ? c : R::get_o_plain(io.frec, c); };
  auto valid = [&](uint32_t c) { return c == DSA_INVALID || ((c >> 2) < F && (c & 3u) != 3u); };
  const uint32_t NVT = io.NV;
  // one candidate: the face across corner oc has its three vertices decoded before entry p?  (ids out)
  auto swing = [&](uint32_t c, bool left) {     // SwingLeft = Next(Opposite(Next(c))), SwingRight = Previous(Opposite(Previous(c)))
    const uint32_t o2 = opposite(left ? qnext(c) : qprev(c));
    if (o2 == DSA_INVALID) return o2;
    if (!valid(o2)) { bad = true; return (uint32_t)DSA_INVALID; }
    return left ? qnext(o2) : qprev(o2);
  };
  for (uint32_t p = tid; p < entries; p += stride) {
    // Two phases, so that the dependent loads of an entry are the walk around its vertex alone (a record per step) and everything
    // the candidates need -- the face across, the entries of its three vertices -- is requested for all of them at once: the walk
    // collects the corners across (up to MP_FAN of them: the fan of nearly every vertex; a longer one is walked the plain way).
    // The triples go straight to memory, the first word of each -- which carries the count -- at the end: a record held in a local
    // array indexed by the count would live in scratch or LDS.
    uint32_t *rec = &prep[p].id[0][0];
    uint32_t first0 = 0, first1 = 0, first2 = 0, first3 = 0;
    const uint32_t start = io.d2c[p];
    uint32_t found = 0;
    auto take = [&](uint32_t en, uint32_t ep, uint32_t eo) {
      first0 = found == 0 ? en : first0; first1 = found == 1 ? en : first1;
      first2 = found == 2 ? en : first2; first3 = found == 3 ? en : first3;
      rec[3 * found + 1] = ep; rec[3 * found + 2] = eo;
      ++found;
    };
    if ((start >> 2) >= F || (start & 3u) == 3u) bad = true;
    else if (p > 0) {
      constexpr int MP_FAN = 8;
      uint32_t cand[MP_FAN];
      uint32_t c = start;
      bool first_pass = true, done = false;
#pragma unroll
      for (int k = 0; k < MP_FAN; ++k) {
        cand[k] = DSA_INVALID;
        if (!done) {
          const uint32_t oc = opposite(c);
          if (oc != DSA_INVALID && !valid(oc)) { bad = true; done = true; }
          else {
            cand[k] = oc;
            c = swing(c, first_pass);
            if (c == start) done = true;
            else if (c == DSA_INVALID) {
              if (first_pass && !bad) { first_pass = false; c = swing(start, false); if (c == DSA_INVALID) done = true; }
              else done = true;
            }
          }
        }
      }
      if (done) {
        uint32_t vv[MP_FAN][3];
#pragma unroll
        for (int k = 0; k < MP_FAN; ++k) {
          const uint32_t oc = cand[k] != DSA_INVALID ? cand[k] : start;            // (a corner that is there)
          vv[k][0] = R::get_v(io.frec, qnext(oc)); vv[k][1] = R::get_v(io.frec, qprev(oc)); vv[k][2] = R::get_v(io.frec, oc);
        }
        int32_t ee[MP_FAN][3];
#pragma unroll
        for (int k = 0; k < MP_FAN; ++k)
#pragma unroll
          for (int j = 0; j < 3; ++j) ee[k][j] = io.v2d[vv[k][j] < NVT ? vv[k][j] : 0u];
#pragma unroll
        for (int k = 0; k < MP_FAN; ++k) {
          const bool ok = cand[k] != DSA_INVALID && vv[k][0] < NVT && vv[k][1] < NVT && vv[k][2] < NVT && ee[k][0] >= 0 && ee[k][1] >= 0 && ee[k][2] >= 0 &&
                          (uint32_t)ee[k][0] < p && (uint32_t)ee[k][1] < p && (uint32_t)ee[k][2] < p;
          if (ok && found < 4) take((uint32_t)ee[k][0], (uint32_t)ee[k][1], (uint32_t)ee[k][2]);
        }
      } else if (!bad) {
        // a fan of more than MP_FAN corners: the reference's loop as it is written
        uint32_t steps = 0;
        c = start; first_pass = true;
        while (c != DSA_INVALID) {
          if (++steps > max_steps || !valid(c)) { bad = true; break; }
          const uint32_t oc = opposite(c);
          if (oc != DSA_INVALID) {
            if (!valid(oc)) { bad = true; break; }
            const uint32_t vo = R::get_v(io.frec, oc), vn = R::get_v(io.frec, qnext(oc)), vp = R::get_v(io.frec, qprev(oc));
            if (vo < NVT && vn < NVT && vp < NVT) {
              const int32_t eo = io.v2d[vo], en = io.v2d[vn], ep = io.v2d[vp];
              if (eo >= 0 && en >= 0 && ep >= 0 && (uint32_t)eo < p && (uint32_t)en < p && (uint32_t)ep < p) {
                take((uint32_t)en, (uint32_t)ep, (uint32_t)eo);
                if (found == 4) break;
              }
            }
          }
          c = swing(c, first_pass);
          if (c == start) break;
          if (c == DSA_INVALID && first_pass && !bad) { first_pass = false; c = swing(start, false); }
        }
      }
    }
    const uint32_t tag = found << MP_FOUND_SHIFT;
    rec[0] = first0 | tag; rec[3] = first1 | tag; rec[6] = first2 | tag; rec[9] = first3 | tag;
    // (the operand words of triples that were not found stay as they are: the chain reads them through a mask of their own)
  }
  if (bad) fail(D, ST_INVALID, 675);
}
__global__ __launch_bounds__(256) void k_multipara_prepare(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n) {
  const uint32_t mesh = blockIdx.y, ai = blockIdx.z;
  if (mesh >= n) return;
  MeshDesc *D = &descs[mesh];
  if (status_of(D) != ST_OK || D->general || ai >= D->num_attributes) return;
  const AttrDesc &a = D->att[ai];
  const MeshLayout &L = layouts[mesh];
  if (!a.have_scheme || a.pred_kind != 4 || a.source == SRC_BYTES || a.num_entries == 0) return;
  if (a.num_entries > L.cap_vertices) { fail(D, ST_INVALID, 675); return; }
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  if (L.rec_compact) multipara_prepare_entries<true>(arena, L, D, ai, tid, stride);
  else multipara_prepare_entries<false>(arena, L, D, ai, tid, stride);
}

// The chain.  Lane l of a wave: attribute slot l >> 4 (four meshes to a wave), component c = (l >> 2) & 3, parallelogram i = l & 3
// (the four parallelograms of a component in one quad: their sum is two DPP adds).
// Per entry: the lane's triple was requested MP_AHEAD entries ago; half that far ahead, with the record there, the correction,
// the flag word of the lane's parallelogram (its place follows from the counts of the entries before it) and the three operands as
// memory holds them -- final for every entry further back than the window, which is all they are used for.  Operands inside the
// window (the last MP_WIN results of the attribute, in LDS) are read when the entry's turn comes.  No load stands in a branch.
#define MP_WIN 256u
#define MP_AHEAD 16      // records requested this many entries ahead (the unrolled loop's length)
#define MP_REST 8        // correction, flag word and operands from memory: this many ahead -- a request takes about as long as eight
                         // entries, and a wave has 63 under way at most (16 x 1 + 8 x 5)
__device__ __forceinline__ uint32_t mp_quad_sum(uint32_t x) {
  x += dpp_mov<0xB1>(x);            // quad_perm [1, 0, 3, 2]
  x += dpp_mov<0x4E>(x);            // quad_perm [2, 3, 0, 1]
  return x;
}
__global__ __launch_bounds__(WAVE) void k_multipara(uint8_t *arena, const MeshLayout *layouts, MeshDesc *descs, uint32_t n, BatchGlobals *G) {
  __shared__ int32_t win[4][MP_WIN][4];
  __builtin_amdgcn_s_setprio(3);              // long chains: issue ahead of whatever shares the SIMD
  const uint32_t lane = lane_id(), slot = lane >> 4, comp = (lane >> 2) & 3u, pi = lane & 3u, ai = blockIdx.y;
  const uint32_t mesh = blockIdx.x * 4u + slot;
  // (lanes without work run along on entry 0 of a mesh that is there, and store nothing)
  bool mine = mesh < n;
  MeshDesc *D = &descs[mine ? mesh : 0];
  mine = mine && status_of(D) == ST_OK && !D->general && ai < D->num_attributes;
  const AttrDesc &a = D->att[mine ? ai : 0];
  const MeshLayout &L = layouts[mine ? mesh : 0];
  mine = mine && a.have_scheme && a.pred_kind == 4 && a.source != SRC_BYTES && a.num_entries != 0 && a.num_entries <= L.cap_vertices && L.tc[ai] != 0;
  if (!__ballot(mine)) return;
  const uint32_t nc = mine ? a.nc_portable : 1u;
  const uint32_t entries = mine ? a.num_entries : 0u;
  uint32_t most = entries;
  for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)most, d, WAVE); most = o > most ? o : most; }
  most = uni(most);
  const uint32_t cc = comp < nc ? comp : 0u;                 // the component this lane reads (lanes beyond the last one shadow component 0)
  // where this lane's results go: its component of the attribute (a lane beyond the last component shadows component 0, value and all)
  int32_t *wst = mine ? (int32_t *)(arena + L.work[ai]) + cc : (int32_t *)&G->pad;
  const uint32_t wstride = mine ? nc : 0u;
  const MpPrep *prep = (const MpPrep *)(arena + (mine ? L.tc[ai] : L.stream));
  int32_t *w = (int32_t *)(arena + (mine ? L.work[ai] : L.stream));
  const uint32_t *cbits = mine ? mp_crease_words_of(arena, L, ai, L.cap_vertices) : (const uint32_t *)(arena + L.stream);
  uint32_t cbase[4], cpos[4], cnum[4];
  for (uint32_t k = 0; k < 4; ++k) { cbase[k] = mine ? mp_crease_base(a, k) : 0u; cpos[k] = 0; cnum[k] = mine ? a.num_crease[k] : 0u; }
  const int32_t mn = mine ? a.wrap_min : 0, mx = mine ? a.wrap_max : 0, max_dif = 1 + mx - mn;
  const uint32_t lastp = entries ? entries - 1u : 0u;
  bool ran_out = false;
  // rings of what was requested: [e % MP_AHEAD]
  uint32_t rid0[MP_AHEAD], rid1[MP_AHEAD], rid2[MP_AHEAD], rword[MP_AHEAD], rbit[MP_AHEAD];
  int32_t rcorr[MP_AHEAD], rg0[MP_AHEAD], rg1[MP_AHEAD], rg2[MP_AHEAD];
  auto request_record = [&](uint32_t e, int s) {             // stage A: the triple of this lane's parallelogram
    const uint32_t q = e <= lastp ? e : lastp;
    const uint32_t *t = prep[q].id[pi];
    rid0[s] = t[0]; rid1[s] = t[1]; rid2[s] = t[2];
  };
  auto request_rest = [&](uint32_t e, int s) {               // stage B: the record is here -- the place of this lane's flag and its word; the operands from memory
    const uint32_t found = e <= lastp ? rid0[s] >> MP_FOUND_SHIFT : 0u;
    const uint32_t ctx = found ? found - 1u : 0u;
    uint32_t at = 0, nn = 0, base = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) if (k == ctx) { at = cpos[k]; nn = cnum[k]; base = cbase[k]; }
    const uint32_t bit = at + pi;
    ran_out = ran_out || (found != 0 && at + found > nn);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) if (k == ctx) cpos[k] += found;
    rbit[s] = bit & 31u;
    rword[s] = cbits[base + (bit < nn ? bit >> 5 : 0u)];
    rcorr[s] = w[(size_t)(e <= lastp ? e : lastp) * nc + cc];
    const uint32_t en = rid0[s] & MP_ID_MASK, ep = rid1[s], eo = rid2[s];       // (triples that were not found hold anything: clamped, and never used)
    rg0[s] = w[(size_t)(en <= lastp ? en : lastp) * nc + cc];
    rg1[s] = w[(size_t)(ep <= lastp ? ep : lastp) * nc + cc];
    rg2[s] = w[(size_t)(eo <= lastp ? eo : lastp) * nc + cc];
  };
#pragma unroll
  for (int s = 0; s < MP_AHEAD; ++s) request_record((uint32_t)s, s);
#pragma unroll
  for (int s = 0; s < MP_REST; ++s) request_rest((uint32_t)s, s);
  int32_t o1 = 0;
  int32_t (*mywin)[4] = win[slot];
  // The window reads of an entry are issued one entry early -- before the entry in front of it is finished -- so that the LDS round
  // trip lies beside that entry's arithmetic instead of in the chain; what they cannot have seen, the result of the entry directly
  // in front (the usual case for one operand: the strip's previous entry), comes from the register it was left in.
  int32_t ln = 0, lp = 0, lo = 0;                            // (entry 0 has no operands)
  for (uint32_t g0 = 0; g0 < most; g0 += MP_AHEAD) {
#pragma unroll
    for (int s = 0; s < MP_AHEAD; ++s) {
      const uint32_t p = g0 + (uint32_t)s;
      const bool live = p <= lastp && entries != 0;
      // the window reads of entry p + 1 (its record arrived long ago: it was requested MP_AHEAD entries before its turn)
      const int sn = (s + 1) % MP_AHEAD;
      const uint32_t en1 = rid0[sn] & MP_ID_MASK, ep1 = rid1[sn], eo1 = rid2[sn];
      const int32_t ln1 = mywin[en1 & (MP_WIN - 1u)][cc], lp1 = mywin[ep1 & (MP_WIN - 1u)][cc], lo1 = mywin[eo1 & (MP_WIN - 1u)][cc];
      const uint32_t found = rid0[s] >> MP_FOUND_SHIFT;
      const uint32_t en = rid0[s] & MP_ID_MASK, ep = rid1[s], eo = rid2[s];
      const bool crease = ((rword[s] >> rbit[s]) & 1u) != 0;
      const bool used = pi < found && !crease && live;
      // the entry in front: the register; near: the window (memory may not have it yet); the rest: memory (requested four entries
      // ago, written hundreds ago)
      const int32_t vn = p - en == 1u ? o1 : (p - en > MP_WIN ? rg0[s] : ln);
      const int32_t vp = p - ep == 1u ? o1 : (p - ep > MP_WIN ? rg1[s] : lp);
      const int32_t vo = p - eo == 1u ? o1 : (p - eo > MP_WIN ? rg2[s] : lo);
      const uint32_t sum = mp_quad_sum(used ? (uint32_t)vn + (uint32_t)vp - (uint32_t)vo : 0u);
      const uint32_t cnt = mp_quad_sum(used ? 1u : 0u);
      // the average, truncated toward zero (the C# operator on the wrapped sum)
      const int32_t x = (int32_t)sum;
      int32_t q = x;
      if (cnt == 2u) q = (x + (int32_t)((uint32_t)x >> 31)) >> 1;
      else if (cnt == 3u) q = (int32_t)(((int64_t)x * 0x55555556ll) >> 32) + (int32_t)((uint32_t)x >> 31);
      else if (cnt == 4u) q = (x + ((x >> 31) & 3)) >> 2;
      const int32_t pred = cnt ? q : (p ? o1 : 0);
      int32_t o = wrap_original(pred, rcorr[s], mn, mx, max_dif);
      o = live ? o : o1;
      // (no store stands in a branch either: a join where the paths differ in what is under way makes the compiler wait for
      // everything.  The four lanes of a quad hold the same result and write it to the same place; behind the last entry that
      // entry is written again; lanes without an attribute write to a word of the batch's globals)
      wst[(size_t)(live ? p : lastp) * wstride] = o;
      mywin[p & (MP_WIN - 1u)][comp] = o;
      o1 = o;
      ln = ln1; lp = lp1; lo = lo1;
      // the ring slot is free: the record MP_AHEAD entries on; and the rest for the entry half that far on, whose record is here
      request_record(p + MP_AHEAD, s);
      request_rest(p + MP_REST, (s + MP_REST) % MP_AHEAD);
    }
  }
  if (__ballot(ran_out && mine)) { if (ran_out && mine && pi == 0 && comp == 0) fail(D, ST_INVALID, 676); }
}
#undef MP_WIN
#undef MP_AHEAD
#undef MP_REST

}  // namespace dsa

namespace dsa {
// k_pack_output: the packed block of a compact download (dsa_batch_download_compact) -- faces narrowed to uint16 where every point
// id of the mesh fits, one point map per distinct map (CompactMesh).  Behind the decode, on the download stream.
__global__ __launch_bounds__(256) void k_pack_output(const uint8_t *arena, const MeshLayout *layouts, const MeshDesc *descs, uint32_t n, const CompactMesh *table,
                                                     uint8_t *packed) {
  const uint32_t mesh = blockIdx.y;
  if (mesh >= n) return;
  const MeshDesc *D = &descs[mesh];
  if (D->status != ST_OK) return;
  const MeshLayout &L = layouts[mesh];
  const CompactMesh &c = table[mesh];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  const uint32_t nc = 3u * D->num_faces, npts = D->num_points;
  const int32_t *faces = (const int32_t *)(arena + L.faces);
  if (c.u16) {
    uint32_t *dst = (uint32_t *)(packed + c.faces);                  // two corners to a word
    for (uint32_t w = tid; w < (nc + 1) / 2; w += stride) {
      const uint32_t lo = (uint32_t)faces[2 * w], hi = 2 * w + 1 < nc ? (uint32_t)faces[2 * w + 1] : 0u;
      dst[w] = (lo & 0xFFFFu) | (hi << 16);
    }
  } else {
    int32_t *dst = (int32_t *)(packed + c.faces);
    for (uint32_t i = tid; i < nc; i += stride) dst[i] = faces[i];
  }
  for (uint32_t a = 0; a < D->num_attributes && a < DSA_MAX_ATT; ++a) {
    if (c.map[a] == ~0ull) continue;
    bool rep = true;
    for (uint32_t k = 0; k < a; ++k) rep = rep && c.map[k] != c.map[a];
    if (!rep) continue;
    const uint32_t *src = (const uint32_t *)(arena + L.map[a]);
    uint32_t *dst = (uint32_t *)(packed + c.map[a]);
    for (uint32_t p = tid; p < npts; p += stride) dst[p] = src[p];
  }
}
}  // namespace dsa
