/*
 * oracle.h — C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY. This is a from-scratch CPU restatement of the
 * reference's algorithm for the HashAggregation / HashBuild / HashProbe path
 * (each function in oracle.cpp cites the reference file:line it follows). It is
 * the checker for the HIP path and the "Velox-algorithm CPU restatement" timed
 * as cpu_baseline. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it. Nothing under velox_amd/ links, imports or
 * calls it.
 *
 * Parity status: folly::hasher<T> is third-party arithmetic (folly
 * v2026.01.05.00, not under /root/reference). The restatement of
 * twang_mix64 / jenkins_rev_mix32 / twang_32from64 is pinned against folly's own
 * known-answer vectors (folly/hash/test/HashTest.cpp: TWang_Mix64,
 * TWang_32From64, Jenkins_Rev_Mix32) in tests/test_oracle_hash.py. The
 * reference itself cannot be compiled here (needs folly, xsimd, boost, fmt,
 * glog ...), so there is no oracle/_ref binary; everything above the hash
 * primitives is pinned by transcribing the reference's formula-style test
 * expectations (VectorHasherTest.cpp, HashTableTest.cpp).
 *
 * Data descriptors (vx355_column / vx355_batch / vx355_out_column and the
 * spec structs) are shared with include/vx355.h so the same inputs can be fed
 * to both sides. All pointers are host pointers here.
 */
#ifndef VX355_ORACLE_H_
#define VX355_ORACLE_H_

#include "../include/vx355.h"

#ifdef __cplusplus
extern "C" {
#endif

/* hash primitives (Appendix D of SURVEY.md; folly/hash/Hash.h) */
uint64_t orc_twang_mix64(uint64_t key);
uint32_t orc_twang_32from64(uint64_t key);
uint32_t orc_jenkins_rev_mix32(uint32_t key);
uint64_t orc_hash_mix(uint64_t upper, uint64_t lower);
uint32_t orc_crc32c_u64(uint32_t checksum, uint64_t value);
uint64_t orc_hash_bytes(uint64_t seed, const char* data, size_t size);
uint32_t orc_xxh32_u32(uint32_t value, uint32_t seed);
/* folly::hasher<T> / NaNAwareHash / StringView hash of ONE value of kind. */
uint64_t orc_hash_value(int32_t type_kind, const void* value);

/* VectorHasher::hash over n_keys columns (same contract as vx355_hash_columns). */
int orc_hash_columns(const vx355_batch* batch, const int32_t* key_cols, int32_t n_keys,
                     const uint64_t* rows, int32_t mix_first, uint64_t* out);

/* VectorHasher object, for transcribing exec/tests/VectorHasherTest.cpp. */
typedef struct orc_hasher orc_hasher;
orc_hasher* orc_hasher_create(int32_t type_kind);
void orc_hasher_destroy(orc_hasher* h);
/* computeValueIds: returns 1 if all selected rows were mappable. */
int orc_hasher_compute_value_ids(orc_hasher* h, const vx355_column* col, int32_t num_rows,
                                 const uint64_t* rows, uint64_t* result);
/* lookupValueIds: clears unmappable rows in rows_inout (must not be NULL). */
void orc_hasher_lookup_value_ids(const orc_hasher* h, const vx355_column* col, int32_t num_rows,
                                 uint64_t* rows_inout, uint64_t* result);
void orc_hasher_cardinality(orc_hasher* h, int32_t reserve_pct, uint64_t* as_range,
                            uint64_t* as_distinct);
uint64_t orc_hasher_enable_value_range(orc_hasher* h, uint64_t multiplier, int32_t reserve_pct);
uint64_t orc_hasher_enable_value_ids(orc_hasher* h, uint64_t multiplier, int32_t reserve_pct);
void orc_hasher_merge(orc_hasher* h, const orc_hasher* other, uint64_t max_num_distinct);
typedef struct orc_hasher_state {
  int32_t is_range, has_range, range_overflow, distinct_overflow;
  int64_t min, max;
  uint64_t multiplier, range_size, num_distinct;
} orc_hasher_state;
void orc_hasher_get_state(const orc_hasher* h, orc_hasher_state* out);

/* Same contracts as the vx355_* standalone kernels. */
int orc_value_ids(const vx355_batch* batch, const int32_t* key_cols,
                  const vx355_value_id_spec* specs, int32_t n_keys, const uint64_t* rows,
                  int32_t lookup, uint64_t* result, uint64_t* rows_out, int32_t* all_mapped);
int orc_filter_compact(const uint64_t* values, const uint64_t* nulls, const uint64_t* rows,
                       int32_t num_rows, int32_t* idx_out, int32_t* n_out);
int orc_partition(const uint64_t* hashes, int32_t num_rows, int32_t kind, int32_t num_partitions,
                  int32_t bit_begin, int32_t bit_end, uint32_t* partitions_out);

/* PrestoPage writer (oracle/presto_page.h): vx355_presto_serialize with host buffers. */
int orc_presto_serialize(const vx355_batch* batch, const int32_t* rows, const int64_t* offsets, int32_t num_pages,
                         int32_t flags, void* out, int64_t out_capacity, int64_t* page_offsets);

/* FilterProject for the vx355_filter_project expression class: row-at-a-time
 * restatement of exec/FilterProject.cpp:102-275 + exec/OperatorUtils.cpp:231-257. */
int orc_filter_project(const vx355_batch* batch, const vx355_filter_term* terms, int32_t n_terms,
                       const vx355_projection* projections, int32_t n_projections, int32_t* idx_out,
                       int32_t* n_out, double* const* proj_out, uint64_t* const* proj_nulls_out);

/* HashAggregation. hash_adaptivity = 0 forces kHash (GroupingSet.cpp:494-496). */
typedef struct orc_agg orc_agg;
int orc_agg_create(const vx355_agg_spec* spec, int32_t hash_adaptivity, orc_agg** out);
int orc_agg_add_input(orc_agg* h, const vx355_batch* batch);
int orc_agg_no_more_input(orc_agg* h);
int orc_agg_get_output(orc_agg* h, vx355_out_column* cols, int32_t num_cols, int32_t max_rows,
                       int32_t* n_out, int32_t* finished);
int orc_agg_get_stats(const orc_agg* h, vx355_agg_stats* out);
void orc_agg_destroy(orc_agg* h);
const char* orc_last_error(void);
/* sum(BIGINT) overflow rule of every aggregation created afterwards: 0 = the reference's
 * (checkedPlus on the running sum in input order, vector/AggregationHook.h:126-135), 1 = the
 * exact total must fit int64 (what libvx355 implements; see oracle.cpp gSumOverflowRule). */
void orc_set_sum_overflow_rule(int32_t rule);

/* HashBuild / HashProbe. */
typedef struct orc_join_build orc_join_build;
typedef struct orc_join_table orc_join_table;
typedef struct orc_join_probe orc_join_probe;
int orc_join_build_create(const vx355_join_build_spec* spec, orc_join_build** out);
int orc_join_build_add_input(orc_join_build* h, const vx355_batch* batch);
int orc_join_build_finish(orc_join_build* h, orc_join_build* const* others, int32_t num_others,
                          orc_join_table** out);
void orc_join_build_destroy(orc_join_build* h);
/* HashTable::parallelJoinBuild (exec/HashTable.cpp:1003-1203) with n threads for tables finished afterwards (bench.py's multi-thread CPU leg; default 1 = serial). */
void orc_set_join_build_threads(int32_t n);
void orc_join_table_release(orc_join_table* t);
int orc_join_table_get_stats(const orc_join_table* t, vx355_join_table_stats* out);
int orc_join_probe_create(orc_join_table* t, const vx355_join_probe_spec* spec,
                          orc_join_probe** out);
/* HashJoinNode::filter as vx355_join_filter_term conjunction (HashProbe::evalFilter). The batch given
 * to add_input must stay alive until its output is drained. */
int orc_join_probe_set_filter(orc_join_probe* h, const vx355_join_filter_term* terms, int32_t n_terms);
int orc_join_probe_add_input(orc_join_probe* h, const vx355_batch* batch);
int orc_join_probe_get_output(orc_join_probe* h, int32_t max_rows, int32_t* mapping_out,
                              int32_t* build_rows_out, vx355_out_column* build_cols,
                              const int32_t* build_col_ids, int32_t num_build_cols,
                              int32_t* n_out, int32_t* finished);
/* HashProbe::getBuildSideOutput: right / full (not probed rows), right semi filter (probed rows). */
int orc_join_probe_get_build_side_output(orc_join_probe* h, int32_t max_rows, int32_t* build_rows_out,
                                         vx355_out_column* build_cols, const int32_t* build_col_ids,
                                         int32_t num_build_cols, int32_t* n_out, int32_t* finished);
void orc_join_probe_destroy(orc_join_probe* h);

/* SplitBlockBloomFilter (common/base/SplitBlockBloomFilter.h:26-129, .cpp:27-34) as used by
 * BigintValuesUsingBloomFilter (type/Filter.h:1294-1360): hash = folly::hasher<int64_t>.
 * 'lanes' = xsimd::batch<uint32_t>::size of the host the reference was built for (8 or 4). */
int64_t orc_bloom_num_blocks(int64_t num_elements, double false_positive, int32_t lanes);
void orc_bloom_insert(uint32_t* blocks, int64_t num_blocks, int32_t lanes, const int64_t* values, int64_t n);
void orc_bloom_test(const uint32_t* blocks, int64_t num_blocks, int32_t lanes, const int64_t* values, int64_t n,
                    uint8_t* may_contain_out);

#ifdef __cplusplus
}
#endif
#endif
