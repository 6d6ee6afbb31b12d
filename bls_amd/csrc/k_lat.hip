// k_lat.hip -- the LATENCY path: one tuple (pairing, Verify, hash tail, scalar multiplication ...) per 64-lane wave
// (programs: gen_lat.py).
//
// The throughput kernels give each tuple one lane pair; a call then costs what one lane pair needs for the whole path
// (~15 ms) however few tuples it carries, and the Go API is one tuple per call (g2pubs/bls.go:159-162).  Here a tuple owns
// a WAVE: its field elements live in LDS slots (15 x 27-bit limbs + 1 pad word = 64 bytes, Montgomery R = 2^405, the
// representation of fp.cuh) and the wave interprets a straight-line program of levels.  In a level every lane does the
// same thing to its own job:
//   MUL  gather two small signed combinations of slots (ds_read_b128), one Montgomery product (the column-scanning core of
//        fp.cuh as MAD chains with rotating carry-out pairs), store the result slot;
//   SQR  one gather, three times its square (the squaring core: the cyclotomic squarings of the final exponentiation);
//   LIN  gather one longer combination -- spread over up to four lanes when the level has few jobs -- and normalise it;
//   INV / SEL / LOAD  an inversion, a table entry picked by a digit of the tuple's scalar, an input element.
// The 54 Fq products of an Fq12 multiplication are one level; the doubling step of the NEXT Miller iteration shares levels
// with the accumulator update of the current one.  MillerLoop + FinalExponentiation: ~510 product / squaring levels +
// ~810 recombination levels instead of ~14 600 sequential multiplications.
//
// Replaces, for small calls: MillerLoop (pairing.go:16-75), FinalExponentiation (pairing.go:79-129), CompareTwoPairings
// (pairing.go:140-147), the curve arithmetic of hash-to-curve (hash.go:185-389, g2.go:104-138), the subgroup tests
// (g1.go:137-141, g2.go:293-295), MulFR (g1.go:80-90) and the tails of the MSM and of VerifyAggregate.  Results leave as
// canonical values (verdict byte, the reference's in-memory FQ12, affine wire bytes), so they are bit-identical to the
// throughput path's and the reference's.
#include "tower.cuh"
#include "device_io.cuh"

namespace {
constexpr int K_MUL = 0, K_LIN = 1, K_INV = 2, K_LOAD = 3, K_OUT12 = 4, K_CHECK1 = 5, K_OUTRAW12 = 6, K_OUTAFF = 7, K_ISZERO = 8, K_SEL = 9, K_SQR = 10;
constexpr int K_REP = 11;               // not a level: "the next `len` (bits 8-15) levels run `count` (bits 16-23) times" -- the rolled squaring runs of gen_lat.py (Pairing.exp_by_x)
constexpr int SLOT_WORDS = 20;              // 15 limbs + pad, at a stride of 80 bytes: see lat_lds_bytes (blsmi.hip)

struct LatHeader {                      // gen_lat.py: encode()
    u32 magic, nlevels, nslot, nconst, out_kind, nout, nchk, r2;   // OUTAFF: nout result elements, then nchk values that must not be zero
    unsigned short out_slot[12];
    u8 pad[8];
};

// x += c * S[slot] over the 15 limbs (the 16th word of a slot is padding); c in [-15, 15].  The accumulators are 64 bits
// wide so that a term costs ONE v_mad_u64_u32 per limb (a 32-bit multiply-add does not exist on gfx950; v_mul_lo + v_add
// would be two); only their low words are meaningful and used.
struct Acc { u64 v[NL]; };
// INIT: the first term of a gather -- the accumulators are written, not added to (saves clearing 15 register pairs)
template <bool INIT>
BLSMI_DEV void gather_term(const i32* S, u32 term, Acc& x) {
    const u32 c = (u32)((i32)(term >> 11) - 16);
    const int4* p = reinterpret_cast<const int4*>(S + (term & 0x7ffu) * SLOT_WORDS);
    const int4 a = p[0], b = p[1], d = p[2], e = p[3];
    const i32 v[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w, e.x, e.y, e.z, e.w};
    // rotating carry-out pairs: see gen_lat_mul.py
    if constexpr (INIT) {
        asm volatile("v_mad_u64_u32 %0, s[36:37], %8, %9, 0\n\tv_mad_u64_u32 %1, s[38:39], %8, %10, 0\n\tv_mad_u64_u32 %2, s[40:41], %8, %11, 0\n\tv_mad_u64_u32 %3, s[42:43], %8, %12, 0\n\t"
                     "v_mad_u64_u32 %4, s[44:45], %8, %13, 0\n\tv_mad_u64_u32 %5, s[46:47], %8, %14, 0\n\tv_mad_u64_u32 %6, s[48:49], %8, %15, 0\n\tv_mad_u64_u32 %7, s[50:51], %8, %16, 0"
                     : "=&v"(x.v[0]), "=&v"(x.v[1]), "=&v"(x.v[2]), "=&v"(x.v[3]), "=&v"(x.v[4]), "=&v"(x.v[5]), "=&v"(x.v[6]), "=&v"(x.v[7])
                     : "v"(c), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7])
                     : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51");
        asm volatile("v_mad_u64_u32 %0, s[36:37], %7, %8, 0\n\tv_mad_u64_u32 %1, s[38:39], %7, %9, 0\n\tv_mad_u64_u32 %2, s[40:41], %7, %10, 0\n\tv_mad_u64_u32 %3, s[42:43], %7, %11, 0\n\t"
                     "v_mad_u64_u32 %4, s[44:45], %7, %12, 0\n\tv_mad_u64_u32 %5, s[46:47], %7, %13, 0\n\tv_mad_u64_u32 %6, s[48:49], %7, %14, 0"
                     : "=&v"(x.v[8]), "=&v"(x.v[9]), "=&v"(x.v[10]), "=&v"(x.v[11]), "=&v"(x.v[12]), "=&v"(x.v[13]), "=&v"(x.v[14])
                     : "v"(c), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14])
                     : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49");
        return;
    }
    asm volatile("v_mad_u64_u32 %0, s[36:37], %8, %9, %0\n\tv_mad_u64_u32 %1, s[38:39], %8, %10, %1\n\tv_mad_u64_u32 %2, s[40:41], %8, %11, %2\n\tv_mad_u64_u32 %3, s[42:43], %8, %12, %3\n\t"
                 "v_mad_u64_u32 %4, s[44:45], %8, %13, %4\n\tv_mad_u64_u32 %5, s[46:47], %8, %14, %5\n\tv_mad_u64_u32 %6, s[48:49], %8, %15, %6\n\tv_mad_u64_u32 %7, s[50:51], %8, %16, %7"
                 : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]), "+v"(x.v[7])
                 : "v"(c), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7])
                 : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51");
    asm volatile("v_mad_u64_u32 %0, s[36:37], %7, %8, %0\n\tv_mad_u64_u32 %1, s[38:39], %7, %9, %1\n\tv_mad_u64_u32 %2, s[40:41], %7, %10, %2\n\tv_mad_u64_u32 %3, s[42:43], %7, %11, %3\n\t"
                 "v_mad_u64_u32 %4, s[44:45], %7, %12, %4\n\tv_mad_u64_u32 %5, s[46:47], %7, %13, %5\n\tv_mad_u64_u32 %6, s[48:49], %7, %14, %6"
                 : "+v"(x.v[8]), "+v"(x.v[9]), "+v"(x.v[10]), "+v"(x.v[11]), "+v"(x.v[12]), "+v"(x.v[13]), "+v"(x.v[14])
                 : "v"(c), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14])
                 : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49");
}
// FIRST: x is written (zero when there is no term), otherwise added to
template <bool FIRST>
BLSMI_DEV void gather(const i32* S, const u32* terms, int nt, Acc& x) {
    if (FIRST) {
        if (nt == 0) {
#pragma unroll
            for (int i = 0; i < NL; i++) x.v[i] = 0;
            return;
        }
        gather_term<true>(S, terms[0], x);
    }
#pragma unroll
    for (int t = FIRST ? 1 : 0; t < 7; t++) {
        if (t >= nt) break;                                                // nt is uniform across the wave (level header)
        gather_term<false>(S, terms[t], x);
    }
}
BLSMI_DEV void acc_zero(Acc& x) {
#pragma unroll
    for (int i = 0; i < NL; i++) x.v[i] = 0;
}
BLSMI_DEV void acc_low(const Acc& x, i32 r[NL]) {
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = (i32)(u32)x.v[i];
}
// Montgomery product for a wave that is ALONE on its SIMD: MAD chains with a rotating carry-out SGPR pair (gen_lat_mul.py).
#include "lat_mul.inc"
BLSMI_DEV void store_slot(i32* S, u32 slot, const i32 r[NL]) {
    int4* p = reinterpret_cast<int4*>(S + slot * SLOT_WORDS);
    p[0] = make_int4(r[0], r[1], r[2], r[3]); p[1] = make_int4(r[4], r[5], r[6], r[7]);
    p[2] = make_int4(r[8], r[9], r[10], r[11]); p[3] = make_int4(r[12], r[13], r[14], 0);
}
}  // namespace

// bufs: up to four input arrays of affine records (48-byte big-endian field elements), stride 0 = one broadcast record.
// out_kind CHECK1: ok[t] = (result == 1) && !flags[t];  OUT12: out[t] = the 12 Fq of the result as Montgomery-384 words;
// OUTRAW12: out = int32 structure-of-arrays buffer of the n results in the device representation.
// ISZERO: ok[t] = the result elements are all zero.
// OUTAFF: out = n records of nout big-endian 48-byte field elements (an affine point), ok[t] = 0 when a step was exceptional.
__global__ void __launch_bounds__(64, 1) k_lat(const u8* prog, const u8* b0, size_t s0, const u8* b1, size_t s1, const u8* b2, size_t s2,
                                                const u8* b3, size_t s3, const u8* flags, u8* ok, u64* out, size_t n) {
    extern __shared__ int4 lds4[];
    i32* S = reinterpret_cast<i32*>(lds4);
    const size_t t = blockIdx.x;
    if (t >= n) return;
    const int lane = threadIdx.x;
    const LatHeader* H = reinterpret_cast<const LatHeader*>(prog);
    const u32 nlevels = H->nlevels, nconst = H->nconst;
    const u32* lvl = reinterpret_cast<const u32*>(prog + sizeof(LatHeader));
    const i32* consts = reinterpret_cast<const i32*>(prog + sizeof(LatHeader) + ((nlevels * 4 + 15) & ~15u));
    const uint4* desc = reinterpret_cast<const uint4*>(consts + (size_t)nconst * 16);
    for (u32 k = lane; k < nconst * 16; k += 64) {                           // constants into their slots (word 15 of a record = its slot)
        const u32 rec = k >> 4, w = k & 15;
        if (w < 15) S[(u32)consts[rec * 16 + 15] * SLOT_WORDS + w] = consts[k];
    }
    __syncthreads();
    uint4 d0 = desc[(size_t)lane * 2], d1 = desc[(size_t)lane * 2 + 1];
    // l walks the header words, dl the descriptor blocks (one per LEVEL: a K_REP word has none).  A loop is entered at its K_REP word and
    // left after `count` passes over [rep_lo, rep_hi); all of this is wave-uniform scalar state.
    u32 dl = 0, rep_left = 0, rep_lo = 0, rep_hi = 0xffffffffu, rep_dlo = 0;
    for (u32 l = 0; l < nlevels;) {
        const u32 h = lvl[l];
        if ((h & 0x7f) == K_REP) {                                             // (the prefetched descriptors are the loop's first level's: dl does not move)
            rep_left = (h >> 16) & 0xff; rep_lo = l + 1; rep_hi = l + 1 + ((h >> 8) & 0xff); rep_dlo = dl;
            l++;
            continue;
        }
        u32 nl = l + 1, ndl = dl + 1;                                          // where the wave goes after this level
        if (nl == rep_hi) {
            if (rep_left > 1) { rep_left--; nl = rep_lo; ndl = rep_dlo; }
            else rep_hi = 0xffffffffu;
        }
        const int kind = h & 0x7f, ntx = (h >> 8) & 0xff, nty = (h >> 16) & 0xff, njobs = (int)(h >> 24);
        const bool reduce = (h & 0x80) != 0;
        const u32 f[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};     // 16 x u16: dst, 7 x-terms, 7 y-terms, flags
        u32 tx[7], ty[7];
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const int jx = 1 + i, jy = 8 + i;
            tx[i] = (f[jx >> 1] >> (16 * (jx & 1))) & 0xffffu;
            ty[i] = (f[jy >> 1] >> (16 * (jy & 1))) & 0xffffu;
        }
        const u32 dst = f[0] & 0xffffu;
        if (nl < nlevels) {                                                    // next level's descriptors while this one computes
            d0 = desc[((size_t)ndl * 64 + lane) * 2]; d1 = desc[((size_t)ndl * 64 + lane) * 2 + 1];
        }
        i32 r[NL];
        if (kind == K_MUL) {
            Acc ax, ay;
            gather<true>(S, tx, ntx, ax);
            gather<true>(S, ty, nty, ay);
            i32 x[NL], y[NL];
            acc_low(ax, x); acc_low(ay, y);
            lat_mul(x, y, r);
        } else if (kind == K_SQR) {                                            // 3 x^2: the cyclotomic squarings (gen_lat.py cyc_sqr)
            Acc ax;
            gather<true>(S, tx, ntx, ax);
            i32 x[NL];
            acc_low(ax, x);
            lat_sqr3(x, r);
        } else if (kind == K_LIN) {
            Acc ax;
            gather<true>(S, tx, ntx, ax);
            gather<false>(S, ty, nty, ax);
            Fp<LMAX, VMAX> x;
            acc_low(ax, x.v);
            // njobs of a LIN level = lanes per job (gen_lat.py encode): the job's terms are spread over 2 or 4 adjacent lanes,
            // whose partial sums meet here (quad_perm [1,0,3,2], then [2,3,0,1]); every lane of the group ends with the total
            if (njobs >= 2) {
#pragma unroll
                for (int i = 0; i < NL; i++) x.v[i] += __builtin_amdgcn_update_dpp(0, x.v[i], 0xB1, 0xf, 0xf, true);
            }
            if (njobs == 4) {
#pragma unroll
                for (int i = 0; i < NL; i++) x.v[i] += __builtin_amdgcn_update_dpp(0, x.v[i], 0x4E, 0xf, 0xf, true);
            }
            if (reduce) { const Fp<1, 3> y = fp_reduce(x); for (int i = 0; i < NL; i++) r[i] = y.v[i]; }
            else { const auto y = fp_norm(x); for (int i = 0; i < NL; i++) r[i] = y.v[i]; }
        } else if (kind == K_INV) {
            Acc ax;
            gather<true>(S, tx, ntx, ax);
            Fp<LMAX, VMAX> x;
            acc_low(ax, x.v);
            const FpS y = fp_inv(x);                                           // inverse(0) = 0
#pragma unroll
            for (int i = 0; i < NL; i++) r[i] = y.v[i];
        } else if (kind == K_SEL) {                                            // table entry picked by a 4-bit digit of the tuple's digit record (buffer 1, big-endian bytes)
            // raw fields: tx[0] = slot of entry 0, tx[1] = window (0 = least significant), ty[0] = slots per entry, ty[1] = index of the record's last byte
#pragma unroll
            for (int i = 0; i < NL; i++) r[i] = 0;
            if (lane < njobs) {
                const u32 w = tx[1];
                const u32 digit = ((u32)(b1 + s1 * t)[ty[1] - (w >> 1)] >> ((w & 1u) * 4u)) & 15u;
                const int4* p = reinterpret_cast<const int4*>(S + (tx[0] + digit * ty[0]) * SLOT_WORDS);
                const int4 a = p[0], b = p[1], d = p[2], e = p[3];
                r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
                r[8] = d.x; r[9] = d.y; r[10] = d.z; r[11] = d.w; r[12] = e.x; r[13] = e.y; r[14] = e.z;
            }
        } else {                                                               // K_LOAD: tx[0] = buffer | element << 4
            const u32 bsel = tx[0] & 15u, el = (tx[0] >> 4) & 0xfffu;
            FpS y = fp_zero();
            if (lane < njobs) {
                if (bsel == 8 || bsel == 14) {                                     // device representation: raw limbs, SoA with n = 1, at buffer 3 (8) / buffer 2 (14)
                    const i32* raw = reinterpret_cast<const i32*>(bsel == 8 ? b3 + s3 * t : b2 + s2 * t) + el * NL;
#pragma unroll
                    for (int i = 0; i < NL; i++) y.v[i] = raw[i];
                } else if (bsel == 12 || bsel == 13) {                              // a coordinate of a projective point: 12 = from the affine wire format, 13 = SoA raw
                    const u32 e = el & 15u, which = (el >> 4) & 1u, six = (el >> 5) & 1u;
                    const u32 ycoord = six ? 2u : 1u, zcoord = six ? 4u : 2u;
                    const size_t idx = t + (which ? s2 : 0), cnt = s3;
                    bool inf = idx >= cnt;
                    if (bsel == 12) {
                        const u8* rec = b0 + s0 * (inf ? 0 : idx);
                        if (!inf) {
                            const u32* w32 = reinterpret_cast<const u32*>(rec);
                            u32 any = 0;
                            for (size_t i = 0; i < s0 / 4; i++) any |= w32[i];
                            inf = any == 0 || (b1 && b1[idx]);                     // the all-zero record / the caller's flag: the point at infinity
                        }
                        if (!inf) { if (e < zcoord) y = load_be48(rec + 48 * e); else if (e == zcoord) y = C_ONE; }
                    } else if (!inf) {
                        const i32* raw = reinterpret_cast<const i32*>(b3);
#pragma unroll
                        for (int i = 0; i < NL; i++) y.v[i] = raw[((size_t)e * NL + i) * cnt + idx];
                    }
                    if (inf && e == ycoord) y = C_ONE;                             // (0 : 1 : 0)
                } else if (bsel == 11) {                                           // Fq12 product tree: element e of SoA record t + which * s2 of buffer 3 (s3 records); past the end: 1
                    const u32 e = el & 15u, which = (el >> 4) & 1u;
                    const size_t idx = t + (which ? s2 : 0), cnt = s3;
                    if (idx < cnt) {
                        const i32* raw = reinterpret_cast<const i32*>(b3);
#pragma unroll
                        for (int i = 0; i < NL; i++) y.v[i] = raw[((size_t)e * NL + i) * cnt + idx];
                    } else if (e == 0) y = C_ONE;
                } else if (bsel == 10) {                                           // coordinate of a Jacobian SoA record (msm.inc, curve.cuh jac_soa_store) at buffer 3
                    const u32 e = el & 7u, w = (el >> 3) & 255u, ncoord = (el >> 11) & 1u ? 6u : 3u;
                    const i32* raw = reinterpret_cast<const i32*>(b3);
                    const size_t cnt = s3;                                         // records in the buffer = word stride
#pragma unroll
                    for (int i = 0; i < NL; i++) y.v[i] = raw[((size_t)e * NL + i) * cnt + w];
                    if (raw[(size_t)ncoord * NL * cnt + w]) {                      // flagged infinite: (0, 1, 0) whatever the coordinates hold
                        y = fp_zero();
                        if (e == ncoord / 3) y = C_ONE;
                    }
                } else if (bsel == 9) {                                            // Fq wire format (Montgomery 2^384 limbs) at buffer 0
                    y = load_m384(reinterpret_cast<const u64*>(b0 + s0 * t) + 6 * el);
                } else {
                    const u8* base = bsel == 0 ? b0 + s0 * t : bsel == 1 ? b1 + s1 * t : bsel == 2 ? b2 + s2 * t : b3 + s3 * t;
                    y = load_be48(base + 48 * el);
                }
            }
#pragma unroll
            for (int i = 0; i < NL; i++) r[i] = y.v[i];
        }
        // One wave per workgroup: LDS operations of a wave execute in program order, so the gathers above precede these
        // stores and the stores precede the next level's gathers without a barrier (a barrier would also wait for the
        // descriptor prefetch).  The wave barrier only pins the compiler's ordering; it emits no instruction.
        __builtin_amdgcn_wave_barrier();
        store_slot(S, dst, r);
        __builtin_amdgcn_wave_barrier();
        l = nl; dl = ndl;
    }
    // results leave LDS
    const int okind = (int)H->out_kind;
    FpS v = fp_zero();
    const int nres = (okind == K_OUTAFF || okind == K_ISZERO) ? (int)(H->nout + H->nchk) : okind == K_OUTRAW12 ? (int)H->nout : 12;
    if (lane < nres) {
        const i32* p = S + (u32)H->out_slot[lane] * SLOT_WORDS;
#pragma unroll
        for (int i = 0; i < NL; i++) v.v[i] = p[i];
    }
    if (okind == K_CHECK1) {
        const bool good = lane >= 12 ? true : (lane == 0 ? fp_eq(v, C_ONE) : fp_is_zero(v));
        const bool all = __all(good ? 1 : 0) != 0;
        if (lane == 0) ok[t] = (all && !(flags && flags[t])) ? 1 : 0;
    } else if (okind == K_ISZERO) {                                            // ok[t] = every result element is zero
        const bool nz = lane < nres && !fp_is_zero(v);
        const bool any_nz = __any(nz ? 1 : 0) != 0;
        if (lane == 0) ok[t] = any_nz ? 0 : 1;
    } else if (okind == K_OUTAFF) {                                            // affine coordinates in the wire format; ok[t] = no check value is zero
        const int nout = (int)H->nout;
        const bool bad = lane >= nout && lane < nres && fp_is_zero(v);
        const bool any_bad = __any(bad ? 1 : 0) != 0;
        if (lane < nout) store_be48(reinterpret_cast<u8*>(out) + 48 * ((size_t)nout * t + lane), v);
        if (lane == 0 && ok) ok[t] = any_bad ? 0 : 1;
    } else if (okind == K_OUTRAW12) {                                          // device representation, SoA over the n tuples (the product tree's input)
        if (lane < nres) {                                                     // nout elements (12 for an Fq12, 3 / 6 for a projective point)
            i32* fbuf = reinterpret_cast<i32*>(out);
#pragma unroll
            for (int j = 0; j < NL; j++) fbuf[((size_t)lane * NL + j) * n + t] = v.v[j];
        }
    } else if (lane < 12) {
        store_m384(out + 72 * t + 6 * lane, v);
    }
}
