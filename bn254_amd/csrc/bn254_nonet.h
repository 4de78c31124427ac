// The NONET layout of the final exponentiation (bn254_nonet.hip): ONE verify carried by nine lane pairs — the nine squarings of a
// Granger-Scott squaring at once, the 18 products of a Karatsuba Fq12 product in two rounds of nine, everything linear distributed as
// well — written ONCE against a "box" (where the exchanged values live) so that the SAME source is
//   * the kernel k_final_exp_nonet (box = the workgroup's LDS, references = word offsets of the lane's verify and role), and
//   * a host emulation (box = plain arrays, the nine pairs run one after the other per exchange step; tests/hostsim:
//     hp_nonet_check) that the CPU suite runs against fe_machine on every verify case and, built with -DBN_TRACK_BOUNDS, under the
//     interval tracker — the proof that no 64-bit column, int32 limb or value bound can be exceeded in THIS arrangement of the pair
//     layout's formulas (tests/test_pair_layout.py::test_nonet_schedule_*).
// Include after bn254_pairing.h (pair layout: BN_SPLIT_FP2).
//
// Every select below is on a constant of the lane's POSITION (so_odd, l1_xi_w, ...), never on data: fp2_select_pos — the tracker follows the
// operand that position takes, and the emulation runs all nine positions (bn254_field.h: fp_select_pos).
//
// Structure of an operation: every pair computes a value from what the box holds ("phase function", no side effect), then the pairs
// PUBLISH (write) — on the device between wavefront-scope fences inside one wave (a wave's LDS instructions execute in order), on the
// host as two loops over the pairs.  Which coefficient / product / output a pair handles is DATA (references worked out once by
// nn_lane_roles from the index tables below), never control flow.
//
// INVARIANT of the replicated operations (LOAD, STORE, CONJ, FROB, INV): they are executed by ALL lanes of a verify (and by the follow-along
// lanes 54..63 of a wave, on the wave's last verify) on the verify's single shared accumulator / slot file; every lane of a role computes
// bit-identical words from bit-identical inputs and the lanes run in lockstep (one wave, no lane-dependent control flow in those routines),
// so the concurrent read-modify-writes of one address all store the same word.  A routine that gains a lane-dependent path breaks this.
#pragma once

namespace bn254 {

#define BN_NONET_PAIRS 9
#if defined(__HIPCC__)
#define NN_TABLE __device__ __constant__ const unsigned char
#else
#define NN_TABLE static const unsigned char
#endif
#if defined(BN_TRACK_BOUNDS) && !defined(__HIPCC__)
#define NN_MODE const int          // BN_SITE_MODE reads the run-time table of the site search
#else
#define NN_MODE constexpr int
#endif

NN_TABLE NN_SQ_I1[9] = {0, 4, 0, 3, 2, 3, 1, 5, 1};        // pair 3k + j squares x (j = 0), y (j = 1), carry(x + y) (j = 2)
NN_TABLE NN_SQ_I2[9] = {6, 6, 4, 6, 6, 2, 6, 6, 5};        // of group k: (c0.c0, c1.c1), (c1.c0, c0.c2), (c0.c1, c1.c2); 6 = zero
NN_TABLE NN_SQ_OUT_K[6] = {0, 1, 2, 2, 0, 1};              // output coefficient w: its group ...
NN_TABLE NN_SQ_OUT_ODD[6] = {0, 0, 0, 1, 1, 1};            // ... and whether it is the odd (2xy) half
// product q = 6 g + K: group g = 0: t0 = a.c0 b.c0, 1: t1 = a.c1 b.c1, 2: u = carry(a.c0 + a.c1) carry(b.c0 + b.c1); Karatsuba operand
// K = 0, 1, 2: coefficient K; 3: c1 + c2; 4: c0 + c1; 5: c0 + c2.  An operand is carry(C[i1] + C[i2]) + carry(C[i3] + C[i4]) with
// 6 = zero: for g < 2 the carries act on tight values (no-ops in value, uniform code), for g = 2 they are sites 28..33.
NN_TABLE NN_MUL_IDX[18][4] = {
    {0, 6, 6, 6}, {1, 6, 6, 6}, {2, 6, 6, 6}, {1, 6, 2, 6}, {0, 6, 1, 6}, {0, 6, 2, 6},
    {3, 6, 6, 6}, {4, 6, 6, 6}, {5, 6, 6, 6}, {4, 6, 5, 6}, {3, 6, 4, 6}, {3, 6, 5, 6},
    {0, 3, 6, 6}, {1, 4, 6, 6}, {2, 5, 6, 6}, {1, 4, 2, 5}, {0, 3, 1, 4}, {0, 3, 2, 5}};
// Fq6 coefficient K of group g from its four products (fp6_kfin_coef): pk - pA - pB, xi on it for K = 0, + pD (xi on it for K = 1)
NN_TABLE NN_L1_A[3] = {1, 0, 0};
NN_TABLE NN_L1_B[3] = {2, 1, 2};
NN_TABLE NN_L1_D[3] = {0, 2, 1};
// output coefficient w: A + [xi] B + C with B, C negated for the c1 half (u - t0 - t1); 9 = zero
NN_TABLE NN_L2_A[6] = {0, 1, 2, 6, 7, 8};
NN_TABLE NN_L2_B[6] = {5, 3, 4, 0, 1, 2};
NN_TABLE NN_L2_C[6] = {9, 9, 9, 3, 4, 5};

// What a lane pair needs to know about its place, as references into the box (Box::Ref: an LDS word offset on the device, a pointer on the
// host), worked out ONCE and kept in registers: an indexed table read inside an operation would be a global-memory round trip on the
// latency path of a lone wave.
template <class Box> struct NnLane {
  typedef typename Box::Ref Ref;
  typedef typename Box::Rel Rel;
  unsigned pair;                // lane pair within the verify (0..8)
  bool writer;                  // the lane belongs to a verify (lanes 54..63 of a wave follow the wave's last verify and publish nothing)
  bool publishes_out;           // ... and its pair forms an output coefficient (pairs 0..5)
  // cyclotomic squaring
  Ref sq_a, sq_b, sq_pub;       // operand = carry(box[sq_a] + box[sq_b]); its square goes to sq_pub
  Ref so_x2, so_y2, so_s2, so_coef;
  bool so_odd, so_xi;
  int32_t so_sign;
  // multiplication: operand coefficient j of round r at (accumulator | slot) + m_rel[r][j], or the zero block
  Rel m_rel[2][4];
  bool m_zero[2][4];
  Ref m_pub[2];
  Ref l1_k, l1_a, l1_b, l1_d, l1_pub;
  bool l1_xi_w, l1_xi_d;
  Ref l2_a, l2_b, l2_c, out_coef;
  bool l2_xi;
  int32_t l2_neg;
};
template <class Box> BN_DEV void nn_lane_roles(NnLane<Box>& L, Box& bx, unsigned p, bool writer) {
  L.pair = p;
  L.writer = writer;
  L.publishes_out = writer && p < 6;
  const unsigned wq = p < 6 ? p : p - 6;                       // pairs 6..8 repeat outputs 0..2 and publish nothing
  L.sq_a = bx.coef(NN_SQ_I1[p]); L.sq_b = bx.coef(NN_SQ_I2[p]); L.sq_pub = bx.xp(p);
  const unsigned k = NN_SQ_OUT_K[wq];
  L.so_odd = NN_SQ_OUT_ODD[wq] != 0; L.so_xi = L.so_odd && k == 2; L.so_sign = L.so_odd ? 2 : -2;
  L.so_x2 = bx.xp(3 * k); L.so_y2 = bx.xp(3 * k + 1); L.so_s2 = bx.xp(3 * k + 2); L.so_coef = bx.coef(wq);
  for (unsigned r = 0; r < 2; ++r) {
    const unsigned q = p + 9 * r;
    for (unsigned j = 0; j < 4; ++j) { const unsigned idx = NN_MUL_IDX[q][j]; L.m_zero[r][j] = idx >= 6; L.m_rel[r][j] = Box::rel(idx < 6 ? idx : 0u); }
    L.m_pub[r] = bx.xp(q);
  }
  const unsigned g = p / 3, K = p % 3;
  L.l1_k = bx.xp(6 * g + 3 + K); L.l1_a = bx.xp(6 * g + NN_L1_A[K]); L.l1_b = bx.xp(6 * g + NN_L1_B[K]); L.l1_d = bx.xp(6 * g + NN_L1_D[K]);
  L.l1_pub = bx.x1(p);
  L.l1_xi_w = K == 0; L.l1_xi_d = K == 1;
  L.l2_a = bx.x1(NN_L2_A[wq]); L.l2_b = bx.x1(NN_L2_B[wq]); L.l2_c = bx.x1(NN_L2_C[wq]); L.out_coef = bx.coef(wq);
  L.l2_xi = wq == 0; L.l2_neg = wq >= 3 ? -1 : 0;
}
// -x where m is all ones, x where m is zero
BN_DEV Fp2 nn_cond_neg(const Fp2& x, int32_t m) {
  Fp2 r;
#if defined(__HIPCC__)
#pragma unroll
  for (int i = 0; i < BN_LIMBS; ++i) r.c[0].v[i] = (x.c[0].v[i] ^ m) - m;
#else
  BN_FOR_ROLES(k) r.c[k] = fp_select_pos(m != 0, fp_neg(x.c[k]), x.c[k]);
#endif
  return r;
}
constexpr int nn_max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

// ---- Granger-Scott squaring of the accumulator (bn254_field.h: fp12_cyclotomic_sqr_body<170>, fp4_sqr<S>) -------------------------------
// Sites: where the three Fq4 squarings of the pair layout differ in a site's mode the strictest is applied to all — a carry or a weak
// reduction changes no value mod q and only tightens limbs.
template <class Box> BN_DEV Fp2 nn_csqr_square(const NnLane<Box>& L, Box& bx) {           // phase 1: the pair's squaring leaf
  NN_MODE m_sum = nn_max3(BN_SITE_MODE(170, 1), BN_SITE_MODE(173, 1), BN_SITE_MODE(176, 1));
  return fp2_sqr(fp2_site(fp2_add(bx.get(L.sq_a), bx.get(L.sq_b)), m_sum));
}
template <class Box> BN_DEV Fp2 nn_csqr_output(const NnLane<Box>& L, Box& bx) {           // phase 2: output coefficient 3t -+ 2a
  NN_MODE m_odd = nn_max3(BN_SITE_MODE(171, 1), BN_SITE_MODE(174, 1), BN_SITE_MODE(177, 1));
  NN_MODE m_even = nn_max3(BN_SITE_MODE(172, 1), BN_SITE_MODE(175, 1), BN_SITE_MODE(178, 1));
  NN_MODE m_xi = BN_SITE_MODE(179, 1);
  const Fp2 x2 = bx.get(L.so_x2), y2 = bx.get(L.so_y2), s2 = bx.get(L.so_s2);
  // even half: x^2 + xi y^2;  odd half: 2xy = s^2 - x^2 - y^2, for c1.c0 times xi — ONE multiplication by xi serves both: of y^2 in the
  // lanes that form an even output, of 2xy in the others (used by the c1.c0 lanes only)
  const Fp2 t_odd = fp2_site(fp2_sub(fp2_sub(s2, x2), y2), m_odd);
  const Fp2 xi_part = fp2_mul_xi(fp2_select_pos(L.so_odd, t_odd, y2));
  // e = x^2 + xi y^2 in the even lanes, xi (2xy) in the odd ones: x^2 is masked out there BEFORE the addition — x^2 + xi (2xy) would be a
  // discarded value in those lanes, but one whose limbs can leave int32 (3 + 4 + 1 units of 2^28 with sites 171 / 174 / 177 off): found by
  // the bound tracker on the host emulation of this schedule (tests/test_pair_layout.py::test_nonet_schedule_bounds_hold)
  const Fp2 e = fp2_site(fp2_add(fp2_select_pos(L.so_odd, fp2_zero(), x2), xi_part), nn_max3(m_even, m_xi, 0));
  return fp2_lin2_reduce(fp2_select_pos(L.so_odd && !L.so_xi, t_odd, e), 3, bx.get(L.so_coef), L.so_sign);
}

// ---- acc <- acc * slot, Karatsuba (bn254_field.h: fp12_mul_body, fp6_mul<S>) ------------------------------------------------------------
template <class Box> BN_DEV Fp2 nn_mul_operand(const NnLane<Box>& L, Box& bx, typename Box::Base base, unsigned r, int mode) {
  Fp2 c[4];
#pragma unroll
  for (unsigned j = 0; j < 4; ++j) c[j] = bx.get(L.m_zero[r][j] ? bx.zero() : Box::at(base, L.m_rel[r][j]));
  return fp2_add(fp2_site(fp2_add(c[0], c[1]), mode), fp2_site(fp2_add(c[2], c[3]), mode));
}
template <class Box> BN_DEV Fp2 nn_mul_product(const NnLane<Box>& L, Box& bx, typename Box::Base a, typename Box::Base b, unsigned r) {   // rounds 0, 1
  NN_MODE m_a = nn_max3(BN_SITE_MODE(28, 1), BN_SITE_MODE(29, 1), BN_SITE_MODE(30, 1));
  NN_MODE m_b = nn_max3(BN_SITE_MODE(31, 1), BN_SITE_MODE(32, 1), BN_SITE_MODE(33, 1));
  return fp2_mul(nn_mul_operand(L, bx, a, r, m_a), nn_mul_operand(L, bx, b, r, m_b));
}
template <class Box> BN_DEV Fp2 nn_mul_level1(const NnLane<Box>& L, Box& bx) {            // pair p forms Fq6 coefficient K = p % 3 of group g = p / 3
  NN_MODE m_in = nn_max3(BN_SITE_MODE(20, 1), BN_SITE_MODE(24, 1), BN_SITE_MODE(34, 1));                 // fp6_mul<S>: NS(S, ...) under the xi
  NN_MODE m_k = nn_max3(nn_max3(BN_SITE_MODE(21, 1), BN_SITE_MODE(25, 1), BN_SITE_MODE(35, 1)), nn_max3(BN_SITE_MODE(22, 1), BN_SITE_MODE(26, 1), BN_SITE_MODE(36, 1)),
                        nn_max3(BN_SITE_MODE(23, 1), BN_SITE_MODE(27, 1), BN_SITE_MODE(37, 1)));
  const Fp2 pk = bx.get(L.l1_k), pa = bx.get(L.l1_a), pb = bx.get(L.l1_b), pd = bx.get(L.l1_d);
  const Fp2 wv = fp2_sub(fp2_sub(pk, pa), pb);
  // K = 0: xi (pk - p1 - p2) + p0;  K = 1: (pk - p0 - p1) + xi p2;  K = 2: (pk - p0 - p2) + p1 — one multiplication by xi, of the bracket or of pD
  const Fp2 xi_part = fp2_mul_xi(fp2_select_pos(L.l1_xi_w, fp2_site(wv, m_in), pd));
  return fp2_site(fp2_add(fp2_select_pos(L.l1_xi_w, xi_part, wv), fp2_select_pos(L.l1_xi_w, pd, fp2_select_pos(L.l1_xi_d, xi_part, pd))), m_k);
}
template <class Box> BN_DEV Fp2 nn_mul_level2(const NnLane<Box>& L, Box& bx) {            // pair w < 6 forms output coefficient w
  NN_MODE m_out = nn_max3(nn_max3(BN_SITE_MODE(38, 2), BN_SITE_MODE(39, 2), BN_SITE_MODE(40, 2)), nn_max3(BN_SITE_MODE(41, 2), BN_SITE_MODE(42, 2), BN_SITE_MODE(43, 2)), 0);
  // c0.cK = t0.cK + (v t1).cK,  c1.cK = u.cK - t0.cK - t1.cK
  const Fp2 a = bx.get(L.l2_a), b = bx.get(L.l2_b), c = bx.get(L.l2_c);
  const Fp2 bs = nn_cond_neg(fp2_select_pos(L.l2_xi, fp2_mul_xi(b), b), L.l2_neg), cs = nn_cond_neg(c, L.l2_neg);
  return fp2_site(fp2_add(fp2_add(a, bs), cs), m_out);
}

#if !defined(__HIPCC__)
// ---- host emulation: the box as plain arrays, the nine pairs of an exchange step one after the other ----------------------------------------
struct NnHostBox {
  typedef Fp2* Ref;
  typedef unsigned Rel;
  typedef Fp12* Base;
  Fp12 acc, file[BN_FE_CHECK_SLOTS];
  Fp2 xpv[18], x1v[9], zerov;
  NnHostBox() { zerov = fp2_zero(); }
  static Fp2* c12(Fp12* x, unsigned idx) { Fp2* c[6] = {&x->c0.c0, &x->c0.c1, &x->c0.c2, &x->c1.c0, &x->c1.c1, &x->c1.c2}; return c[idx]; }
  Ref coef(unsigned idx) { return idx < 6 ? c12(&acc, idx) : &zerov; }
  Ref xp(unsigned q) { return &xpv[q]; }
  Ref x1(unsigned q) { return q < 9 ? &x1v[q] : &zerov; }
  Ref zero() { return &zerov; }
  static Rel rel(unsigned idx) { return idx; }
  static Ref at(Base b, Rel r) { return c12(b, r); }
  Fp2 get(Ref r) const { return *r; }
  void put(Ref r, const Fp2& v) { *r = v; }
};
// the accumulator machine on program `prog` in the nonet schedule; acc in / out.  The replicated operations run once (every pair would
// compute the same words, see the invariant above); the two distributed ones run their phases over the nine pairs, publishes in between.
inline void nn_machine_model(Fp12& acc, const unsigned char (*prog)[2]) {
  static NnHostBox bx;                                           // ~40 KB with the tracker's bookkeeping: not on the stack
  bx.acc = acc;
  static NnLane<NnHostBox> L[BN_NONET_PAIRS];
  for (unsigned p = 0; p < BN_NONET_PAIRS; ++p) nn_lane_roles(L[p], bx, p, true);
  Fp2 t[BN_NONET_PAIRS];
  for (int pc = 0;; ++pc) {
    const int op = prog[pc][0], arg = prog[pc][1];
    if (op == FE_END) break;
    switch (op) {
      case FE_LOAD: bx.acc = bx.file[arg]; break;
      case FE_STORE: bx.file[arg] = bx.acc; break;
      case FE_CSQR:
        for (unsigned p = 0; p < 9; ++p) t[p] = nn_csqr_square(L[p], bx);
        for (unsigned p = 0; p < 9; ++p) bx.put(L[p].sq_pub, t[p]);
        for (unsigned p = 0; p < 9; ++p) t[p] = nn_csqr_output(L[p], bx);
        for (unsigned p = 0; p < 6; ++p) bx.put(L[p].so_coef, t[p]);
        break;
      case FE_MUL:
        for (unsigned r = 0; r < 2; ++r) {
          for (unsigned p = 0; p < 9; ++p) t[p] = nn_mul_product(L[p], bx, &bx.acc, &bx.file[arg], r);
          for (unsigned p = 0; p < 9; ++p) bx.put(L[p].m_pub[r], t[p]);
        }
        for (unsigned p = 0; p < 9; ++p) t[p] = nn_mul_level1(L[p], bx);
        for (unsigned p = 0; p < 9; ++p) bx.put(L[p].l1_pub, t[p]);
        for (unsigned p = 0; p < 9; ++p) t[p] = nn_mul_level2(L[p], bx);
        for (unsigned p = 0; p < 6; ++p) bx.put(L[p].out_coef, t[p]);
        break;
      case FE_CONJ: fp6_neg(bx.acc.c1, bx.acc.c1); break;
      case FE_FROB: fp12_frob_body(bx.acc, bx.acc, arg); break;
      default: fp12_inv(bx.acc, bx.acc); break;
    }
  }
  acc = bx.acc;
}
#endif

}  // namespace bn254
