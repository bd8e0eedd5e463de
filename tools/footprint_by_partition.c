/* developer analysis, see footprint_by_partition.py: the oracle's own traversal with its visit marks, strip by strip */
#include "../oracle/svo_oracle.c"
/* mark the words read by the primary rays of the given 8x8 strips (strip s = block (s % bpr, s / bpr)) */
void fp_mark(const uint32_t *nodes, size_t n_nodes, const oracle_uniforms *u, const uint32_t *strips, size_t n_strips, int bpr, uint32_t *visits) {
    for (size_t k = 0; k < n_strips; k++) {
        frame_job j = {nodes, n_nodes, u, (int)(strips[k] % (uint32_t)bpr) * 8, (int)(strips[k] / (uint32_t)bpr) * 8, 8, 8, NULL, NULL, NULL, visits};
        frame_fn(&j, 0, 64);
    }
}
