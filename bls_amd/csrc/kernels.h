// kernels.h -- declarations of the kernels defined in the k_*.hip translation units, for the host side (blsmi.hip).
// Generated from the definitions by tools/gen_kernel_decls.py; the launch bounds live with the definitions.
#pragma once
#include "fp.cuh"
using namespace blsmi;
#define WG 64
constexpr int PT = WG / 2;          // tuples per workgroup of the lane-pair kernels
constexpr int QT = WG / 4;          // tuples per workgroup of the lane-quad kernels
constexpr int RT = WG / 16;         // tuples per workgroup of the lane-row kernels
// k_pairing_single.hip
__global__ void k_miller1(const u8* g1, const u8* g2, i32* fbuf, size_t n);
__global__ void k_miller1h(const u8* g1, const u8* g2, i32* fbuf, size_t n);
__global__ void k_prepare_generator_lines(const u8* g2, i32* table);
__global__ void k_miller2(const u8* p0, size_t sp0, const u8* q0, size_t sq0, const u8* p1, size_t sp1, const u8* q1, size_t sq1, i32* fbuf, size_t n, const i32* pre);
__global__ void k_debug_prepare_single(const u8* g2, i32* table);
__global__ void k_debug_lines_to_m384(const i32* table, u64* out);
// k_fe_single.hip
__global__ void k_final_exp(const i32* fbuf, u64* out, size_t n, int mode);
__global__ void k_final_exp_is_one(const i32* fbuf, const u8* inf_flags, u8* ok, size_t n);
__global__ void k_final_exp_equal(const i32* a, const i32* b, i32* ok);
// k_fq12_single.hip
__global__ void k_fq12_from_m384(const u64* in, i32* fbuf, size_t n);
__global__ void k_fq12_prod_level(const i32* src, i32* dst, size_t n, size_t half);
__global__ void k_fq12_aos_to_soa(const i32* aos, i32* soa, size_t n);
__global__ void k_fq12_one(i32* f);
__global__ void k_debug_fq12(int op, const u64* a, const u64* b, u64* out, size_t n);
// k_pairing_pair.hip
__global__ void k_debug_pairl(int op, const u64* a, const u64* b, u64* out, size_t n);
__global__ void k_debug_prepare_pair(const u8* g2, i32* table);
__global__ void k_prepare_generator_lines_pair(const u8* g2, i32* table);
// pair_kernels.inc
__global__ void k_miller1_pair(const u8* g1, const u8* g2, i32* fbuf, size_t n);
__global__ void k_miller1h_pair(const u8* g1, const u8* g2, i32* fbuf, size_t n);
__global__ void k_final_exp_pair(const i32* fbuf, u64* out, size_t n, int mode);
__global__ void k_miller1x2_pair(const u8* g1, const u8* g2, i32* fbuf, size_t n, size_t m);
__global__ void k_miller2_pair(const u8* p0, size_t sp0, const u8* q0, size_t sq0, const u8* p1, size_t sp1, const u8* q1, size_t sq1, i32* fbuf, size_t n, const i32* pre);
__global__ void k_final_exp_is_one_pair(const i32* fbuf, const u8* inf_flags, u8* ok, size_t n);
// k_pairing_quad.hip
__global__ void k_miller1h_quad(const u8* g1, const u8* g2, i32* fbuf, size_t n);
__global__ void k_final_exp_quad(const i32* fbuf, u64* out, size_t n, int mode);
__global__ void k_miller2_quad(const u8* p0, size_t sp0, const u8* q0, size_t sq0, const u8* p1, size_t sp1, const u8* q1, size_t sq1, i32* fbuf, size_t n, const i32* pre);
__global__ void k_miller1x2_quad(const u8* g1, const u8* g2, i32* fbuf, size_t n, size_t m);
__global__ void k_final_exp_is_one_quad(const i32* fbuf, const u8* inf_flags, u8* ok, size_t n);
__global__ void k_debug_quad(int op, const u64* a, const u64* b, u64* out, size_t n);
// k_pairing_row.hip
__global__ void k_miller1h_row(const u8* g1, const u8* g2, i32* fbuf, size_t n);
__global__ void k_final_exp_row(const i32* fbuf, u64* out, size_t n, int mode);
__global__ void k_miller2_row(const u8* p0, size_t sp0, const u8* q0, size_t sq0, const u8* p1, size_t sp1, const u8* q1, size_t sq1, i32* fbuf, size_t n, const i32* pre);
__global__ void k_miller1s_row(const u8* p, size_t sp, const u8* q, size_t sq, i32* fbuf, size_t n, const i32* pre, size_t first, size_t end);
__global__ void k_miller1m_row(const u8* p, size_t sp, const u8* q, size_t sq, i32* fbuf, size_t n);
__global__ void k_final_exp_is_one_row(const i32* fbuf, const u8* inf_flags, u8* ok, size_t n);
__global__ void k_clear_h2_row(const i32* jbuf, u8* good, u8* out, size_t n);
__global__ void k_clear_h2_oct(const i32* jbuf, u8* good, u8* out, size_t n);
__global__ void k_debug_row(int op, const u64* a, const u64* b, u64* out, size_t n);
__global__ void k_clear_h2_quad(const i32* jbuf, u8* good, u8* out, size_t n);            // k_hash_quad.hip
__global__ void k_debug_quad_g2(int op, const u64* a, u64* out, size_t n);
__global__ void k_hash_g1_finish_quad(const u8* pts, u8* good, u8* out, size_t n);
__global__ void k_hash_g1_finish_redo(const u8* pts, const u8* good, u8* out, size_t n);   // k_hash.hip
// k_prepared_pair.hip
__global__ void k_g2_prepare_pair(const u8* g2, i32* tables, size_t n);
__global__ void k_prepared_export(const i32* tables, u64* out, size_t n);
__global__ void k_flag_prepared(const i32* tables, const u32* key_idx, const u8* sigs, int sig_words, const u8* in_flags, u8* flags, int* any_flag, size_t n);
__global__ void k_prepared_gather_keys(const i32* tables, const u32* key_idx, u32* pks, size_t n);
__global__ void k_miller1_prep_pair(const u8* g1, const i32* tables, const u32* key_idx, i32* fbuf, size_t n);
__global__ void k_miller2_prep_pair(const u8* sigs, const u8* h, const i32* tables, const u32* key_idx, i32* fbuf, size_t n, const i32* pre_gen);
__global__ void k_miller1x2_prep_pair(const u8* g1, const i32* tables, const u32* key_idx, i32* fbuf, size_t n, size_t m);
// k_lat.hip
__global__ void k_lat(const u8* prog, const u8* b0, size_t s0, const u8* b1, size_t s1, const u8* b2, size_t s2,
                                                const u8* b3, size_t s3, const u8* flags, u8* ok, u64* out, size_t n);
// k_hash.hip
__global__ void k_hash_g1(const u8* msgs, const u64* off, u8* out, size_t n, int clear, int* special);
__global__ void k_hash_g2(const u8* msgs, const u64* off, u8* out, size_t n);
__global__ void k_hash_g2_domain(const u8* msgs32, const u8* domain, u8* out, size_t n);
__global__ void k_swu_g1_two_lanes(const u8* msgs, const u64* off, u8* pts, size_t n);
__global__ void k_hash_g1_finish(const u8* pts, u8* out, size_t n, int clear, int* special);
__global__ void k_swu_g2_two_lanes(const u8* msgs, const u64* off, u8* pts, size_t n);
__global__ void k_swu_g1_waves(const u8* msgs, const u64* off, u8* pts, size_t n);
__global__ void k_swu_g2_waves(const u8* msgs, const u64* off, u8* pts, size_t n);
__global__ void k_swu_g1_rows(const u8* msgs, const u64* off, u8* pts, size_t n);
__global__ void k_swu_g2_rows(const u8* msgs, const u64* off, u8* pts, size_t n);
__global__ void k_tai_g2_lanes8(const u8* msgs32, const u8* domain, u8* pts, size_t n);
__global__ void k_tai_g2_waves8(const u8* msgs32, const u8* domain, u8* pts, size_t n);
__global__ void k_hash_g1_redo(const u8* msgs, const u64* off, const u8* good, u8* out, size_t n);
__global__ void k_hash_g2_redo(const u8* msgs, const u64* off, const u8* good, u8* out, size_t n);
__global__ void k_tai_g2_wave(const u8* msgs32, const u8* domain, u8* pts, size_t n);
__global__ void k_hash_g2_domain_redo(const u8* msgs32, const u8* domain, const u8* good, u8* out, size_t n);
__global__ void k_write_generators(u8* g1, u8* g2);
// k_wire.hip
__global__ void k_g1_decompress(const u8* in, int check, u8* out, u8* out_inf, u8* err, size_t n);
__global__ void k_g1_decompress_waves(const u8* in, u8* out, u8* out_inf, u8* err, size_t n);
__global__ void k_g2_decompress(const u8* in, int check, u8* out, u8* out_inf, u8* err, size_t n);
__global__ void k_g2_decompress_waves(const u8* in, u8* out, u8* out_inf, u8* err, size_t n);
__global__ void k_apply_subgroup(const u8* in_subgroup, u8* out, int rec_words, const u8* out_inf, u8* err, size_t n);
__global__ void k_merge_flags(const u8* inf_pk, const u8* err_pk, const u8* inf_sig, const u8* err_sig, u8* flags, size_t n);
__global__ void k_flag_zero_records(const u8* pks, int pk_words, const u8* sigs, int sig_words, const u8* in_flags, u8* flags, int* any, size_t n);
__global__ void k_flag_to_byte(const i32* flag, u8* out);
__global__ void k_pack_bitmap(const u8* ok, u8* bitmap, size_t n);
__global__ void k_g1_compress(const u8* pts, const u8* in_inf, u8* out, size_t n);
__global__ void k_g2_compress(const u8* pts, const u8* in_inf, u8* out, size_t n);
__global__ void k_g1_jac_to_affine(const u64* in, u8* out, u8* out_inf, size_t n);
__global__ void k_g2_jac_to_affine(const u64* in, u8* out, u8* out_inf, size_t n);
__global__ void k_affine_to_jac(const u8* in, const void* in_inf, int inf_u8, int group, u64* out, size_t n);
__global__ void k_debug_fq(int op, const u64* a, const u64* b, u64* out, u8* flag, size_t n);
__global__ void k_debug_fq2(int op, const u64* a, const u64* b, u64* out, u8* flag, size_t n);
__global__ void k_debug_swu_g1(const u64* a, u64* out, size_t n);
__global__ void k_debug_swu_g2(const u64* a, u64* out, size_t n);
// k_hash_pair.hip
__global__ void k_hash_g2_pair(const u8* msgs, const u64* off, u8* good, u8* out, size_t n, unsigned redo_every);
__global__ void k_cofac2_pair(const u8* pts, u8* out, size_t n);
__global__ void k_hash_g2_front(const u8* msgs, const u64* off, u8* good, i32* jbuf, size_t n, unsigned redo_every);
// k_curve.hip
__global__ void k_debug_fq6(int op, const u64* a, const u64* b, u64* out, size_t n);
__global__ void k_debug_curve(int op, const u64* a, const u64* b, u64* out, size_t n);
__global__ void k_good_to_flag(const u8* good, i32* flag);
__global__ void k_mul_finish(const u8* good, const u8* pts, size_t pt_stride, int rec_words, u8* out, u8* out_inf, size_t n);
__global__ void k_g1_mul(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n);
__global__ void k_g2_mul(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n);
__global__ void k_glv_recode(const u8* scalars, int group, u8* rec, size_t n);
__global__ void k_g1_mul_glv(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n);
__global__ void k_g2_mul_glv(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n);
__global__ void k_fixed_table_from_wire(const u8* wire, int elems, i32* table, size_t n);
__global__ void k_g1_mul_fixed(const i32* table, const u8* scalars, u8* out, u8* out_inf, size_t n);
__global__ void k_g2_mul_fixed(const i32* table, const u8* scalars, u8* out, u8* out_inf, size_t n);
__global__ void k_g1_mul_fixed_wave(const i32* table, const u8* scalars, u8* out, u8* out_inf, size_t n);
__global__ void k_g2_mul_fixed_wave(const i32* table, const u8* scalars, u8* out, u8* out_inf, size_t n);
__global__ void k_g1_sum0(const u8* pts, const u8* in_inf, i32* buf, size_t n, size_t half);
__global__ void k_g2_sum0(const u8* pts, const u8* in_inf, i32* buf, size_t n, size_t half);
__global__ void k_g1_sum0_jac(const u64* pts, i32* buf, size_t n, size_t half);
__global__ void k_g2_sum0_jac(const u64* pts, i32* buf, size_t n, size_t half);
__global__ void k_g1_sum(const i32* src, i32* dst, size_t n, size_t half);
__global__ void k_g2_sum(const i32* src, i32* dst, size_t n, size_t half);
__global__ void k_g1_sum_final(const i32* src, u8* out, i32* out_inf);
__global__ void k_g2_sum_final(const i32* src, u8* out, i32* out_inf);
__global__ void k_g2_mul_pair(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n);
__global__ void k_g2_mul_glv_pair(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n);
// k_msm_pair.hip
__global__ void k_g2_msm_bucket_pair(const u8* pts, const u32* idx, const u32* offs, const u32* hist, const u32* perm, i32* buckets, size_t n, int c, size_t nb);
__global__ void k_g2_msm_bucket_raw_pair(const i32* raw, const u32* idx, const u32* offs, const u32* hist, const u32* perm, i32* buckets, size_t per_win, size_t nb);
__global__ void k_g2_msm_chunk_pair(const i32* buckets, i32* chunks, int c, int K, size_t nb, size_t nchunks_total);
__global__ void k_g2_msm_fold_pair(const i32* src, i32* dst, size_t seg, size_t half, int nwin);
__global__ void k_g2_msm_chunk2_pair(const i32* buckets, i32* out, int c, int K, size_t nb, size_t nct);
__global__ void k_g2_msm_fold2_pair(const i32* src, i32* dst, int narr, int nwin, size_t len, int io);
// msm.inc
__global__ void k_msm_hist(const u8* scalars, size_t n, int c, int nwin, u32* hist);
__global__ void k_msm_scan(const u32* hist, u32* offs, u32* cursor, int c);
__global__ void k_msm_max(const u32* hist, size_t nb, u32* out);
__global__ void k_msm_scatter(const u8* scalars, size_t n, int c, int nwin, u32* cursor, u32* idx);
__global__ void k_msm_class_hist(const u32* hist, size_t nb, u32* cls);
__global__ void k_msm_class_scan(const u32* cls, u32* class_off, u32* cursor);
__global__ void k_msm_class_scatter(const u32* hist, size_t nb, const u32* class_off, u32* cursor, u32* perm);
__global__ void k_g1_msm_bucket(const u8* pts, const u32* idx, const u32* offs, const u32* hist, const u32* perm, i32* buckets, size_t n, int c, size_t nb);
__global__ void k_g2_msm_bucket(const u8* pts, const u32* idx, const u32* offs, const u32* hist, const u32* perm, i32* buckets, size_t n, int c, size_t nb);
__global__ void k_g1_msm_chunk(const i32* buckets, i32* chunks, int c, int K, size_t nb, size_t nct);
__global__ void k_g2_msm_chunk(const i32* buckets, i32* chunks, int c, int K, size_t nb, size_t nct);
__global__ void k_g1_msm_fold(const i32* src, i32* dst, size_t seg, size_t half, int nwin);
__global__ void k_g2_msm_fold(const i32* src, i32* dst, size_t seg, size_t half, int nwin);
__global__ void k_g1_msm_final(const i32* wins, int nwin, int c, u8* out, i32* out_inf);
__global__ void k_g2_msm_final(const i32* wins, int nwin, int c, u8* out, i32* out_inf);
__global__ void k_msm_recode_g1(const u8* scalars, u8* rec, size_t n);
__global__ void k_msm_rawpts_g1(const u8* pts, i32* raw, size_t n);
__global__ void k_msm_recode_g2(const u8* scalars, u8* rec, size_t n);
__global__ void k_msm_rawpts_g2(const u8* pts, i32* raw, size_t n);
__global__ void k_msm_hist_glv(const u8* rec, size_t n, int nbw, u32* hist);
__global__ void k_msm_scatter_glv(const u8* rec, size_t n, int nbw, int sh, u32* cursor, u32* idx);
__global__ void k_msm_items(const u8* rec, size_t n, int nbw, int sh, u32* keys, u32* items);
__global__ void k_msm_runs(const u32* keys, size_t m, u32 nb, u32* first, u32* last);
__global__ void k_msm_run_lengths(const u32* first, const u32* last, size_t nb, u32* hist, u32* out);
__global__ void k_g1_msm_bucket_raw(const i32* raw, const u32* idx, const u32* offs, const u32* hist, const u32* perm, i32* buckets, size_t per_win, size_t nb);
__global__ void k_g1_msm_chunk2(const i32* buckets, i32* out, int c, int K, size_t nb, size_t nct);
__global__ void k_g2_msm_chunk2(const i32* buckets, i32* out, int c, int K, size_t nb, size_t nct);
__global__ void k_g1_msm_fold2(const i32* src, i32* dst, int narr, int nwin, size_t len, int io);
__global__ void k_g2_msm_fold2(const i32* src, i32* dst, int narr, int nwin, size_t len, int io);
__global__ void k_g1_msm_final2(const i32* recs, int nwin, int m, int logk, int c, u8* out, i32* out_inf);
__global__ void k_g2_msm_final2(const i32* recs, int nwin, int m, int logk, int c, u8* out, i32* out_inf);
