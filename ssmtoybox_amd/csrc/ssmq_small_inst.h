// Instantiation helper for the register-resident moment-transform kernels: each (integrand, D, E, N, SEL) shape gets
// the BQ form, the BQ form with the Student-t process model variance, and the classical centred form.
#pragma once
#include "ssmq_apply_small.h"
#include "ssmq_host.h"

#define SSMQ_SMALL_ONE(F, D, E, N, FORM, TP, SEL, OPT)                                               \
    {F, D, E, N, FORM, TP, SEL, OPT, &ssmq::launch_apply_small<D, E, N, F, FORM, TP, SEL, OPT>,       \
     "k_apply_small<D=" #D ",E=" #E ",N=" #N "," #F "," #FORM ",TP=" #TP ",SEL=" #SEL ",OPT=" #OPT ">"}
#define SSMQ_SMALL(F, D, E, N, SEL)                          \
    SSMQ_SMALL_ONE(F, D, E, N, SSMQ_FORM_BQ, 0, SEL, 0),     \
    SSMQ_SMALL_ONE(F, D, E, N, SSMQ_FORM_BQ, 1, SEL, 0),     \
    SSMQ_SMALL_ONE(F, D, E, N, SSMQ_FORM_SIGMA, 0, SEL, 0)
// larger shapes additionally get the fast paths (SSMQ_OPT_LDL | SSMQ_OPT_UT [| SSMQ_OPT_SYM] for BQ, SSMQ_OPT_UT for the rest)
#define SSMQ_SMALL_FAST(F, D, E, N, SEL)                     \
    SSMQ_SMALL(F, D, E, N, SEL),                             \
    SSMQ_SMALL_ONE(F, D, E, N, SSMQ_FORM_BQ, 0, SEL, 7),     \
    SSMQ_SMALL_ONE(F, D, E, N, SSMQ_FORM_BQ, 0, SEL, 3),     \
    SSMQ_SMALL_ONE(F, D, E, N, SSMQ_FORM_BQ, 0, SEL, 1),     \
    SSMQ_SMALL_ONE(F, D, E, N, SSMQ_FORM_BQ, 1, SEL, 2),     \
    SSMQ_SMALL_ONE(F, D, E, N, SSMQ_FORM_SIGMA, 0, SEL, 2)
