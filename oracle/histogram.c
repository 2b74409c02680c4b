/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see histogram.h).
 * Restates src/utils/histogram.rs of the reference; each function cites the
 * lines it follows.
 */
#include "histogram.h"

#include <stdlib.h>
#include <string.h>

/* histogram.rs:172-178 */
int orc_hist_init(orc_histogram *h, uint64_t capacity) {
    h->values = (uint64_t *)calloc((size_t)capacity + 1, sizeof(uint64_t));
    h->range_start = 0;
    h->range_stop = capacity;
    return h->values ? ORC_OK : ORC_PANIC;
}

/* histogram.rs:394-398 */
int orc_hist_init_default(orc_histogram *h) { return orc_hist_init(h, 512); }

void orc_hist_free(orc_histogram *h) {
    free(h->values);
    h->values = NULL;
}

/* histogram.rs:185-187 */
int orc_hist_increment(orc_histogram *h, uint64_t bin) { return orc_hist_increment_by(h, bin, 1); }

/* histogram.rs:190-197 */
int orc_hist_increment_by(orc_histogram *h, uint64_t bin, uint64_t value) {
    if (bin < h->range_start || bin > h->range_stop) return ORC_BIN_OUT_OF_BOUNDS;
    h->values[bin] += value;
    return ORC_OK;
}

/* histogram.rs:200-207 */
uint64_t orc_hist_get(const orc_histogram *h, uint64_t bin) { return h->values[bin]; }

/* histogram.rs:225-227 */
uint64_t orc_hist_range_len(const orc_histogram *h) { return h->range_stop - h->range_start + 1; }

/* histogram.rs:250-252 */
int orc_hist_in_range(const orc_histogram *h, uint64_t v) {
    return v >= h->range_start && v <= h->range_stop;
}

/* histogram.rs:258-269: sequential f64 accumulation; the product bin_value * i
 * is formed in integers and cast once. */
double orc_hist_mean(const orc_histogram *h) {
    double sum = 0.0;
    double denominator = 0.0;
    for (uint64_t i = h->range_start; i <= h->range_stop; i++) {
        uint64_t bin_value = orc_hist_get(h, i);
        denominator += (double)bin_value;
        sum += (double)(bin_value * i);
    }
    return sum / denominator;
}

/* histogram.rs:272-337 */
int orc_hist_percentile(const orc_histogram *h, double percentile, int *is_some, double *out) {
    *is_some = 0;
    *out = 0.0;
    /* (1) :274-276 */
    if (!(percentile >= 0.0 && percentile <= 1.0)) return ORC_BAD_PERCENTILE;
    /* (2) :279-282 */
    uint64_t num_items = 0;
    for (uint64_t i = h->range_start; i <= h->range_stop; i++) num_items += orc_hist_get(h, i);
    /* (3) :286-288 */
    if (num_items == 0) return ORC_OK;
    /* (4) :292 */
    double needed_items = percentile * (double)num_items;
    /* (5) :297-336 */
    double collected_items = 0.0;
    uint64_t index = h->range_start;
    for (;;) {
        if (index > h->range_stop) return ORC_PANIC; /* :302-304 bail!("Unknown error!") */
        collected_items += (double)orc_hist_get(h, index); /* :308 */
        if (collected_items > needed_items) { /* :312-314 */
            *is_some = 1;
            *out = (double)index;
            return ORC_OK;
        }
        if (collected_items == needed_items) { /* :321-332 */
            uint64_t lowest = index;
            index += 1;
            for (;;) {
                if (index > h->range_stop) return ORC_PANIC; /* get() past the end panics */
                if (orc_hist_get(h, index) != 0) break;
                index += 1;
            }
            uint64_t highest = index;
            *is_some = 1;
            *out = (double)lowest + ((double)(highest - lowest) / 2.0);
            return ORC_OK;
        }
        index += 1; /* :335 */
    }
}

int orc_hist_first_quartile(const orc_histogram *h, int *is_some, double *out) {
    return orc_hist_percentile(h, 0.25, is_some, out);
}
int orc_hist_median(const orc_histogram *h, int *is_some, double *out) {
    return orc_hist_percentile(h, 0.5, is_some, out);
}
int orc_hist_third_quartile(const orc_histogram *h, int *is_some, double *out) {
    return orc_hist_percentile(h, 0.75, is_some, out);
}

/* histogram.rs:355-363 */
int orc_hist_interquartile_range(const orc_histogram *h, int *is_some, double *out) {
    int s1 = 0, s3 = 0;
    double q1 = 0, q3 = 0;
    int rc = orc_hist_first_quartile(h, &s1, &q1);
    if (rc) return rc;
    rc = orc_hist_third_quartile(h, &s3, &q3);
    if (rc) return rc;
    *is_some = s1 && s3;
    *out = (s1 && s3) ? q3 - q1 : 0.0;
    return ORC_OK;
}

/* histogram.rs:366-368 */
uint64_t orc_hist_sum(const orc_histogram *h) {
    uint64_t s = 0;
    for (uint64_t i = h->range_start; i <= h->range_stop; i++) s += h->values[i];
    return s;
}

/* histogram.rs:371-378 */
uint64_t orc_hist_count_from_bottom_until(const orc_histogram *h, uint64_t bin) {
    uint64_t s = 0;
    for (uint64_t i = h->range_start; i <= bin && i <= h->range_stop; i++) s += orc_hist_get(h, i);
    return s;
}

/* histogram.rs:384-391 */
uint64_t orc_hist_count_from_top_until(const orc_histogram *h, uint64_t bin) {
    uint64_t s = 0;
    for (uint64_t i = bin; i <= h->range_stop; i++) s += orc_hist_get(h, i);
    return s;
}

/* histogram.rs:215-218 */
void orc_hist_values_normalized(const orc_histogram *h, double *out) {
    double total = (double)orc_hist_sum(h);
    for (uint64_t i = 0; i <= h->range_stop; i++) out[i] = (double)h->values[i] / total;
}
