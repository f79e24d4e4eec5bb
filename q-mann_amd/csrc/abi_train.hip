// abi_train.hip -- backward / weight-update verbs of boundary B.
//
// The hot path of this library is the test-phase forward.  The
// reference's training verbs are imported by layer.o / MemN2N.o, so the symbols
// must exist for the unmodified host to link; they are outside the scope of
// this round (section 8(f) row 1).  Calling one is a hard error -- message and exit(),
// the boundary's own error convention -- never a silent no-op.
#include "rt.h"
#include "../../include/qmann_abi.h"

#define QM_TRAIN_VERB(name)                                                                   \
    do {                                                                                      \
        fprintf(stderr,                                                                       \
                "[*E] qmann : %s : training verb is not provided by the MI355X inference "    \
                "library (forward path only)\n", name);                                       \
        exit(70);                                                                             \
    } while (0)

extern "C" {

void cuda_dot_mat_vec_bwd(float *, float *, float *, float *, float *, float *, unsigned int, unsigned int, bool,
                          bool, unsigned int, unsigned int, unsigned int, unsigned int, unsigned int, bool)
{ QM_TRAIN_VERB("cuda_dot_mat_vec_bwd"); }

void cuda_dot_mat_vec_bwd_appx(float *, float *, float *, float *, float *, float *, float *, unsigned int,
                               unsigned int, bool, unsigned int, unsigned int, unsigned int, unsigned int, bool,
                               bool, unsigned int)
{ QM_TRAIN_VERB("cuda_dot_mat_vec_bwd_appx"); }

void cuda_softmax_bwd(float *, float *, float *, float *, unsigned int, bool, bool)
{ QM_TRAIN_VERB("cuda_softmax_bwd"); }

void cuda_sum_vec_bwd(float *, float *, float *, float *, unsigned int)
{ QM_TRAIN_VERB("cuda_sum_vec_bwd"); }

void cuda_dense_bwd(float *, float *, float *, float *, float *, float *, float *, float *, float *, unsigned int,
                    unsigned int, char *, bool, unsigned int, unsigned int, unsigned int, unsigned int,
                    unsigned int, bool)
{ QM_TRAIN_VERB("cuda_dense_bwd"); }

void cuda_dense_w_up(float *, float *, float *, float *, float *, float *, unsigned int, unsigned int,
                     unsigned int, float *, float *, float *, bool, unsigned int, unsigned int, unsigned int, bool)
{ QM_TRAIN_VERB("cuda_dense_w_up"); }

void cuda_dense_mat_bwd(float *, float *, float *, float *, float *, float *, float *, float *, unsigned int,
                        unsigned int, unsigned int, bool, unsigned int, unsigned int, unsigned int, bool)
{ QM_TRAIN_VERB("cuda_dense_mat_bwd"); }

void cuda_dense_mat_w_up(float *, float *, float *, float *, float *, float *, float *, float *, unsigned int,
                         unsigned int, unsigned int, float *, float *, float *, bool, unsigned int, unsigned int,
                         unsigned int, bool)
{ QM_TRAIN_VERB("cuda_dense_mat_w_up"); }

void cuda_activation_bwd(float *, float *, float *, char *, unsigned int, bool, unsigned int, unsigned int,
                         unsigned int)
{ QM_TRAIN_VERB("cuda_activation_bwd"); }

void cuda_scale_bwd(float *, float *, float *, float *, float *, unsigned int, bool, unsigned int, unsigned int,
                    unsigned int, bool)
{ QM_TRAIN_VERB("cuda_scale_bwd"); }

void cuda_scale_w_up(float *, float *, unsigned int, unsigned int, float *, float *, bool, unsigned int,
                     unsigned int, unsigned int, bool)
{ QM_TRAIN_VERB("cuda_scale_w_up"); }

void cuda_mult_e_vec_bwd(float *, float *, float *, float *, float *, float *, float *, float *, unsigned int)
{ QM_TRAIN_VERB("cuda_mult_e_vec_bwd"); }

void cuda_mult_e_mat_bwd(float *, float *, float *, float *, float *, float *, float *, float *, unsigned int,
                         unsigned int)
{ QM_TRAIN_VERB("cuda_mult_e_mat_bwd"); }

void cuda_dup_grad_bwd(float *, float *, float *, float *, unsigned int, bool, unsigned int, unsigned int,
                       unsigned int)
{ QM_TRAIN_VERB("cuda_dup_grad_bwd"); }

}  // extern "C"
