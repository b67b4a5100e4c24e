// Register-resident kernels, 5-D reentry vehicle + radar (reference model).
#include "ssmq_small_inst.h"
namespace ssmq {
static const SmallEntry kTable[] = {
    SSMQ_SMALL(SSMQ_F_REENTRY2D_DYN, 5, 5, 10, 0), SSMQ_SMALL_FAST(SSMQ_F_REENTRY2D_DYN, 5, 5, 11, 0),
    SSMQ_SMALL(SSMQ_F_RADAR2D_MEAS, 5, 2, 10, 0), SSMQ_SMALL_FAST(SSMQ_F_RADAR2D_MEAS, 5, 2, 11, 0),
    SSMQ_SMALL(SSMQ_F_RADAR2D_MEAS, 5, 2, 10, 1), SSMQ_SMALL(SSMQ_F_RADAR2D_MEAS, 5, 2, 11, 1),
};
const SmallEntry *small_table_b(int *n) { *n = (int)(sizeof(kTable) / sizeof(kTable[0])); return kTable; }
}  // namespace ssmq
