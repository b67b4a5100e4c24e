// Register-resident kernels, 1- to 3-dimensional models (UNGM, pendulum, reentry-1D): SR (2D), UT/FS-3 (2D+1) and
// small Gauss-Hermite grids.
#include "ssmq_small_inst.h"
namespace ssmq {
static const SmallEntry kTable[] = {
    SSMQ_SMALL(SSMQ_F_UNGM_DYN, 1, 1, 2, 0), SSMQ_SMALL(SSMQ_F_UNGM_DYN, 1, 1, 3, 0),
    SSMQ_SMALL(SSMQ_F_UNGM_DYN, 1, 1, 5, 0), SSMQ_SMALL(SSMQ_F_UNGM_DYN, 1, 1, 7, 0),
    SSMQ_SMALL(SSMQ_F_UNGM_MEAS, 1, 1, 2, 0), SSMQ_SMALL(SSMQ_F_UNGM_MEAS, 1, 1, 3, 0),
    SSMQ_SMALL(SSMQ_F_UNGM_MEAS, 1, 1, 5, 0), SSMQ_SMALL(SSMQ_F_UNGM_MEAS, 1, 1, 7, 0),
    SSMQ_SMALL(SSMQ_F_UNGMNA_DYN, 2, 1, 4, 0), SSMQ_SMALL(SSMQ_F_UNGMNA_DYN, 2, 1, 5, 0),
    SSMQ_SMALL(SSMQ_F_UNGMNA_DYN, 2, 1, 9, 0),
    SSMQ_SMALL(SSMQ_F_UNGMNA_MEAS, 2, 1, 4, 0), SSMQ_SMALL(SSMQ_F_UNGMNA_MEAS, 2, 1, 5, 0),
    SSMQ_SMALL(SSMQ_F_UNGMNA_MEAS, 2, 1, 9, 0),
    SSMQ_SMALL(SSMQ_F_PENDULUM_DYN, 2, 2, 4, 0), SSMQ_SMALL(SSMQ_F_PENDULUM_DYN, 2, 2, 5, 0),
    SSMQ_SMALL(SSMQ_F_PENDULUM_DYN, 2, 2, 9, 0),
    SSMQ_SMALL(SSMQ_F_PENDULUM_MEAS, 2, 1, 4, 0), SSMQ_SMALL(SSMQ_F_PENDULUM_MEAS, 2, 1, 5, 0),
    SSMQ_SMALL(SSMQ_F_PENDULUM_MEAS, 2, 1, 9, 0),
    SSMQ_SMALL(SSMQ_F_REENTRY1D_DYN, 3, 3, 6, 0), SSMQ_SMALL(SSMQ_F_REENTRY1D_DYN, 3, 3, 7, 0),
    SSMQ_SMALL(SSMQ_F_RANGE_MEAS, 3, 1, 6, 0), SSMQ_SMALL(SSMQ_F_RANGE_MEAS, 3, 1, 7, 0),
};
const SmallEntry *small_table_a(int *n) { *n = (int)(sizeof(kTable) / sizeof(kTable[0])); return kTable; }
}  // namespace ssmq
