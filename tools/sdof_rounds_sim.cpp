// Test infrastructure (not product): simulates the "Jacobi rounds to the fixed point" form of the ordered propagation sweeps
// (semi_dense_optical_flow.hpp:146-201) on the CPU, checks that it reproduces the serial sweep bit for bit, and reports how many
// rounds and how many cell recomputations each sweep needs.  Built on the oracle's helpers.
#include "../oracle/oracle.cpp"
#include <set>
#include <cstdio>

extern "C" int sim_sdof_rounds(const vpp_image_desc* i1d, const vpp_image_desc* i2d, const int32_t* kps, int n, int winsize,
                               int nscales, int min_scale, int niters, int patchsize, int32_t* stats /* per sweep: rounds, total work, max list, changes, mismatch */, int max_stats) {
  Img in1(i1d), in2(i2d);
  std::vector<OwnedImg> flow(nscales), mark(nscales), dmap(nscales), P1(nscales), P2(nscales);
  {
    int fr = in1.nr / patchsize, fc = in1.nc / patchsize, ir = in1.nr, ic = in1.nc;
    for (int s = 0; s < nscales; s++) {
      flow[s].alloc(fr, fc, VPP_I32, 2, nscales); mark[s].alloc(fr, fc, VPP_U8, 1, nscales); dmap[s].alloc(fr, fc, VPP_I32, 1, nscales);
      P1[s].alloc(ir, ic, VPP_U8, 1, 2 * winsize); P2[s].alloc(ir, ic, VPP_U8, 1, 2 * winsize);
      fr = 1 + fr / 2; fc = 1 + fc / 2; ir = 1 + ir / 2; ic = 1 + ic / 2;
    }
  }
  auto build = [&](std::vector<OwnedImg>& P, const Img& in) {
    for (int r = 0; r < in.nr; r++) memcpy(P[0].v.row<uint8_t>(r), in.row<uint8_t>(r), in.nc);
    fill_border_generic(P[0].v, 1, VPP_BORDER_MIRROR, nullptr);
    for (int s = 1; s < nscales; s++) pyr_down_t<uint8_t, int>(P[s].v, P[s - 1].v);
  };
  build(P1, in1); build(P2, in2);
  int sweep_no = 0;
  for (int scale = nscales - 1; scale >= min_scale; scale--) {
    int scale_div = 1 << scale;
    const Img i1 = P1[scale].v, i2 = P2[scale].v;
    auto distance = [&](int a0, int a1, int b0, int b1, int max_distance) {
      if (i1.has(a0, a1) && i2.has(b0, b1)) return sad_distance(i1, i2, a0, a1, b0, b1, winsize, max_distance);
      return INT_MAX;
    };
    Img fm = flow[scale].v, mk = mark[scale].v, dm = dmap[scale].v;
    { uint8_t z = 0; vpp_image_desc d{mk.p0, mk.nr, mk.nc, mk.pitch, mk.border, VPP_U8, 1}; orc_fill(&d, &z, 1); }
    for (int i = 0; i < n; i++) {
      int p0 = kps[2 * i] / scale_div, p1 = kps[2 * i + 1] / scale_div;
      int pf0 = p0 / patchsize, pf1 = p1 / patchsize;
      if (!mk.row<uint8_t>(pf0)[pf1]) {
        int pfm0 = p0 / (2 * patchsize), pfm1 = p1 / (2 * patchsize);
        int pr0 = p0, pr1 = p1;
        if (scale < nscales - 1 && mark[scale + 1].v.row<uint8_t>(pfm0)[pfm1]) {
          const int32_t* f = flow[scale + 1].v.row<int32_t>(pfm0) + 2 * pfm1;
          pr0 = p0 + f[0] * 2; pr1 = p1 + f[1] * 2;
        }
        GdMatch m = gradient_descent_match(p0, p1, pr0, pr1, distance, 5);
        int32_t* f = fm.row<int32_t>(pf0) + 2 * pf1;
        f[0] = m.f0; f[1] = m.f1; dm.row<int32_t>(pf0)[pf1] = m.distance; mk.row<uint8_t>(pf0)[pf1] = 2;
      }
    }
    const int NI = (i1.nr - 1) / patchsize + 1, NJ = (i1.nc - 1) / patchsize + 1;
    struct C { int f0, f1, dist, mark; bool operator!=(const C& o) const { return f0 != o.f0 || f1 != o.f1 || dist != o.dist || mark != o.mark; } };
    for (int Ki = 0; Ki < niters; Ki++) {
      const bool forward = Ki % 2;
      // cells by (ci, cj) in map coordinates; map rows visited: forward ci = i, reverse ci = (nr-1 - i*patch)/patch
      auto get = [&](int ci, int cj) { const int32_t* f = fm.row<int32_t>(ci) + 2 * cj; return C{f[0], f[1], dm.row<int32_t>(ci)[cj], mk.row<uint8_t>(ci)[cj]}; };
      std::vector<C> pre((size_t)NI * NJ), S((size_t)NI * NJ);
      for (int i = 0; i < NI; i++) for (int j = 0; j < NJ; j++) pre[(size_t)i * NJ + j] = get(i, j);
      // the visit of wavefront cell (iw, jw) is at image point:
      auto pt = [&](int iw, int jw, int& r, int& c) { if (forward) { r = iw * patchsize; c = jw * patchsize; } else { r = i1.nr - 1 - iw * patchsize; c = i1.nc - 1 - jw * patchsize; } };
      // work in "visit" coordinates (iw, jw) -> map cell (r / patch, c / patch)
      auto cellof = [&](int iw, int jw, int& ci, int& cj) { int r, c; pt(iw, jw, r, c); ci = r / patchsize; cj = c / patchsize; };
      // eval: loop_body for visit (iw, jw) with earlier neighbours from S, later from pre
      long nsad = 0, ngd = 0;
      auto eval = [&](int iw, int jw, const std::vector<C>& Sv) -> C {
        int r, c; pt(iw, jw, r, c);
        const int ci = r / patchsize, cj = c / patchsize;
        C cur = pre[(size_t)ci * NJ + cj];
        if (!cur.mark) return cur;
        const int prev0 = cur.f0, prev1 = cur.f1;
        for (int dr = -1; dr <= 1; dr++)
          for (int dc = -1; dc <= 1; dc++) {
            if (!dr && !dc) continue;
            const int q0 = ci + dr, q1 = cj + dc;
            if (!fm.has(q0, q1)) continue;
            if (q0 >= NI || q1 >= NJ) { fprintf(stderr, "neighbour outside sweep domain\n"); }
            // is (q0, q1) visited earlier than (ci, cj)?  forward: raster order; reverse: reverse raster
            const bool earlier = forward ? (dr < 0 || (dr == 0 && dc < 0)) : (dr > 0 || (dr == 0 && dc > 0));
            const C nb = (q0 < NI && q1 < NJ) ? (earlier ? Sv[(size_t)q0 * NJ + q1] : pre[(size_t)q0 * NJ + q1]) : get(q0, q1);
            if (!nb.mark) continue;
            auto inorm = [](int a, int b) { return (int)std::sqrt((double)(a * a + b * b)); };
            if (inorm(cur.f0 - nb.f0, cur.f1 - nb.f1) > 2 && inorm(prev0 - nb.f0, prev1 - nb.f1) > 2) {
              const int d1 = cur.dist;
              const int d2 = distance(r, c, r + nb.f0, c + nb.f1, INT_MAX); nsad++;
              if (d2 < d1) {
                GdMatch m = gradient_descent_match(r, c, r + nb.f0, c + nb.f1, distance, 5); ngd++;
                if (m.distance < d1) { cur.mark = 1; cur.f0 = m.f0; cur.f1 = m.f1; cur.dist = m.distance; }
              }
            }
          }
        return cur;
      };
      const int NIw = NI, NJw = NJ;
      // round 0
      std::vector<char> changed((size_t)NI * NJ, 0), changed2((size_t)NI * NJ, 0);
      S = pre;
      {
        std::vector<C> S0 = pre;
        for (int iw = 0; iw < NIw; iw++) for (int jw = 0; jw < NJw; jw++) { int ci, cj; cellof(iw, jw, ci, cj); C v = eval(iw, jw, pre); S0[(size_t)ci * NJ + cj] = v; changed[(size_t)ci * NJ + cj] = v != pre[(size_t)ci * NJ + cj]; }
        S = S0;
      }
      long j_changes = 0; for (char ch : changed) j_changes += ch;
      long sad0 = nsad, gd0 = ngd; nsad = ngd = 0;
      int rounds = 0; long total = 0, maxlist = 0, total_changes = 0;
      while (true) {
        // worklist: visits whose earlier neighbours changed in the previous round
        std::vector<std::pair<int, int>> W;
        for (int iw = 0; iw < NIw; iw++) for (int jw = 0; jw < NJw; jw++) {
          int ci, cj; cellof(iw, jw, ci, cj);
          if (!pre[(size_t)ci * NJ + cj].mark) continue;
          bool dirty = false;
          const int s = forward ? -1 : 1;
          const int e[4][2] = {{s, -1}, {s, 0}, {s, 1}, {0, s}};
          for (auto& d : e) { const int q0 = ci + d[0], q1 = cj + d[1]; if (q0 >= 0 && q1 >= 0 && q0 < NI && q1 < NJ && changed[(size_t)q0 * NJ + q1]) dirty = true; }
          if (dirty) W.push_back({iw, jw});
        }
        if (W.empty()) break;
        rounds++; total += (long)W.size(); maxlist = std::max<long>(maxlist, (long)W.size());
        std::vector<C> nv(W.size());
        for (size_t k = 0; k < W.size(); k++) nv[k] = eval(W[k].first, W[k].second, S);
        std::fill(changed2.begin(), changed2.end(), 0);
        for (size_t k = 0; k < W.size(); k++) { int ci, cj; cellof(W[k].first, W[k].second, ci, cj); if (nv[k] != S[(size_t)ci * NJ + cj]) { changed2[(size_t)ci * NJ + cj] = 1; total_changes++; } S[(size_t)ci * NJ + cj] = nv[k]; }
        changed.swap(changed2);
      }
      // serial sweep on the real maps
      auto loop_body = [&](int r, int c) {
        int pf0 = r / patchsize, pf1 = c / patchsize;
        if (!mk.row<uint8_t>(pf0)[pf1]) return;
        int32_t* fpf = fm.row<int32_t>(pf0) + 2 * pf1;
        int prev0 = fpf[0], prev1 = fpf[1];
        for (int dr = -1; dr <= 1; dr++)
          for (int dc = -1; dc <= 1; dc++) {
            if (!dr && !dc) continue;
            int q0 = pf0 + dr, q1 = pf1 + dc;
            if (!(fm.has(q0, q1) && mk.row<uint8_t>(q0)[q1])) continue;
            const int32_t* fq = fm.row<int32_t>(q0) + 2 * q1;
            auto inorm = [](int a, int b) { return (int)std::sqrt((double)(a * a + b * b)); };
            if (inorm(fpf[0] - fq[0], fpf[1] - fq[1]) > 2 && inorm(prev0 - fq[0], prev1 - fq[1]) > 2) {
              int d1 = dm.row<int32_t>(pf0)[pf1];
              int d2 = distance(r, c, r + fq[0], c + fq[1], INT_MAX);
              if (d2 < d1) {
                GdMatch m = gradient_descent_match(r, c, r + fq[0], c + fq[1], distance, 5);
                if (m.distance < d1) { mk.row<uint8_t>(pf0)[pf1] = 1; fpf[0] = m.f0; fpf[1] = m.f1; dm.row<int32_t>(pf0)[pf1] = m.distance; }
              }
            }
          }
      };
      if (forward) { for (int r = 0; r < i1.nr; r += patchsize) for (int c = 0; c < i1.nc; c += patchsize) loop_body(r, c); }
      else { for (int r = i1.nr - 1; r >= 0; r -= patchsize) for (int c = i1.nc - 1; c >= 0; c -= patchsize) loop_body(r, c); }
      long mism = 0;
      for (int i = 0; i < NI; i++) for (int j = 0; j < NJ; j++) if (get(i, j) != S[(size_t)i * NJ + j]) mism++;
      if (sweep_no < max_stats) {
        int32_t* o = stats + 10 * sweep_no;
        o[0] = scale; o[1] = Ki; o[2] = rounds; o[3] = (int)total; o[4] = (int)maxlist; o[5] = (int)j_changes; o[6] = (int)total_changes; o[7] = (int)mism; o[8] = (int)nsad; o[9] = (int)ngd;
      }
      sweep_no++;
    }
  }
  return sweep_no;
}
