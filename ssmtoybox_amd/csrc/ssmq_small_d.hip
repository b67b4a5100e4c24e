// Register-resident kernels, the synthetic 6-D reentry-shaped case of the headline measurement (SURVEY.md 8d, C3:
// D = E = 6, N = 13) and the 7-D non-additive CTRS model.
#include "ssmq_small_inst.h"
namespace ssmq {
static const SmallEntry kTable[] = {
    SSMQ_SMALL(SSMQ_F_REENTRY2D_BIAS_DYN, 6, 6, 12, 0), SSMQ_SMALL_FAST(SSMQ_F_REENTRY2D_BIAS_DYN, 6, 6, 13, 0),
    SSMQ_SMALL(SSMQ_F_RADAR2D_MEAS, 6, 2, 12, 0), SSMQ_SMALL_FAST(SSMQ_F_RADAR2D_MEAS, 6, 2, 13, 0),
    SSMQ_SMALL(SSMQ_F_CTRS_DYN, 7, 5, 14, 0), SSMQ_SMALL(SSMQ_F_CTRS_DYN, 7, 5, 15, 0),
};
const SmallEntry *small_table_d(int *n) { *n = (int)(sizeof(kTable) / sizeof(kTable[0])); return kTable; }
}  // namespace ssmq
