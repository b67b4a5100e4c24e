// Register-resident kernels, coordinated turn / constant velocity + bearing / radar (SURVEY.md config C4 shapes).
#include "ssmq_small_inst.h"
namespace ssmq {
static const SmallEntry kTable[] = {
    SSMQ_SMALL(SSMQ_F_CT_DYN, 5, 5, 10, 0), SSMQ_SMALL_FAST(SSMQ_F_CT_DYN, 5, 5, 11, 0),
    SSMQ_SMALL(SSMQ_F_BEARING_MEAS, 5, 4, 10, 1), SSMQ_SMALL_FAST(SSMQ_F_BEARING_MEAS, 5, 4, 11, 1),
    SSMQ_SMALL(SSMQ_F_BEARING_MEAS, 5, 4, 10, 0), SSMQ_SMALL(SSMQ_F_BEARING_MEAS, 5, 4, 11, 0),
    SSMQ_SMALL(SSMQ_F_CV_DYN, 4, 4, 8, 0), SSMQ_SMALL(SSMQ_F_CV_DYN, 4, 4, 9, 0),
    SSMQ_SMALL(SSMQ_F_RADAR2D_MEAS, 4, 2, 8, 1), SSMQ_SMALL(SSMQ_F_RADAR2D_MEAS, 4, 2, 9, 1),
};
const SmallEntry *small_table_c(int *n) { *n = (int)(sizeof(kTable) / sizeof(kTable[0])); return kTable; }
}  // namespace ssmq
