/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may build, load or run it, and only as the checker.
 *
 * histogram.h/.c: CPU restatement of the reference's Histogram
 * (reference: src/utils/histogram.rs:152-398).  PINNED against the reference's
 * own known-answer tests (src/utils/histogram.rs:405-523 and the doc examples
 * :41-108, :241-248) in tests/test_oracle_kat.py.
 */
#ifndef ORC_HISTOGRAM_H
#define ORC_HISTOGRAM_H

#include <stddef.h>
#include <stdint.h>

/* histogram.rs:152-159  struct Histogram { values: Vec<usize>, range_start, range_stop } */
typedef struct orc_histogram {
    uint64_t *values;
    uint64_t range_start;
    uint64_t range_stop;
} orc_histogram;

#define ORC_OK 0
#define ORC_BIN_OUT_OF_BOUNDS 1 /* histogram.rs:161-164 BinOutOfBoundsError */
#define ORC_PANIC 2             /* the reference would panic (index past the end) */
#define ORC_BAD_PERCENTILE 3    /* histogram.rs:274-276 bail!                    */

int orc_hist_init(orc_histogram *h, uint64_t capacity);       /* :172-178 zero_based_with_capacity */
int orc_hist_init_default(orc_histogram *h);                  /* :394-398 Default = capacity 512   */
void orc_hist_free(orc_histogram *h);
int orc_hist_increment(orc_histogram *h, uint64_t bin);       /* :185-187 */
int orc_hist_increment_by(orc_histogram *h, uint64_t bin, uint64_t v); /* :190-197 */
uint64_t orc_hist_get(const orc_histogram *h, uint64_t bin);  /* :200-207 (caller keeps bin in range) */
uint64_t orc_hist_range_len(const orc_histogram *h);          /* :225-227 */
int orc_hist_in_range(const orc_histogram *h, uint64_t v);    /* :250-252 */
double orc_hist_mean(const orc_histogram *h);                 /* :258-269 */
/* :272-337; *is_some = 0 for Ok(None) */
int orc_hist_percentile(const orc_histogram *h, double p, int *is_some, double *out);
int orc_hist_first_quartile(const orc_histogram *h, int *is_some, double *out); /* :340-342 */
int orc_hist_median(const orc_histogram *h, int *is_some, double *out);         /* :345-347 */
int orc_hist_third_quartile(const orc_histogram *h, int *is_some, double *out); /* :350-352 */
int orc_hist_interquartile_range(const orc_histogram *h, int *is_some, double *out); /* :355-363 */
uint64_t orc_hist_sum(const orc_histogram *h);                           /* :366-368 */
uint64_t orc_hist_count_from_bottom_until(const orc_histogram *h, uint64_t bin); /* :371-378 */
uint64_t orc_hist_count_from_top_until(const orc_histogram *h, uint64_t bin);    /* :384-391 */
void orc_hist_values_normalized(const orc_histogram *h, double *out);    /* :215-218 */

#endif
