/*
 * cuda_stubs.c -- TEST INFRASTRUCTURE ONLY.
 *
 * The 66 `cuda_*` symbols the reference's layer.o (56) and MemN2N.o (10) import
 * (declared for the product in include/qmann_abi.h), as stubs that
 * abort when called.  The reference's CPU code that pins the oracle and serves as
 * the "reference" CPU baseline (oracle/_ref/libqmann_ref*.so, libqmann_refcpu_*.so)
 * links against THIS file, not against the product library: with en_gpu_model =
 * false none of these is ever reached, and the checker's process holds no product
 * code.  layer.c calls them without prototypes, so one `void f(void)` per name
 * satisfies the linker.
 */
#include <stdio.h>
#include <stdlib.h>

#define QM_STUB(name)                                                              \
    void name(void)                                                                \
    {                                                                              \
        fprintf(stderr, "[oracle/cuda_stubs.c] %s called: the CPU-only reference " \
                        "build must run with en_gpu_model = false\n", #name);      \
        abort();                                                                   \
    }

QM_STUB(cuda_accum_mat)
QM_STUB(cuda_activation_bwd)
QM_STUB(cuda_activation_constructor)
QM_STUB(cuda_activation_destructor)
QM_STUB(cuda_activation_fwd)
QM_STUB(cuda_activation_init)
QM_STUB(cuda_copy_dev2host)
QM_STUB(cuda_copy_mat)
QM_STUB(cuda_cross_entropy_constructor)
QM_STUB(cuda_cross_entropy_cost_load)
QM_STUB(cuda_cross_entropy_destructor)
QM_STUB(cuda_cross_entropy_init)
QM_STUB(cuda_cross_entropy_m_cnt_load)
QM_STUB(cuda_cross_entropy_run)
QM_STUB(cuda_data_constructor)
QM_STUB(cuda_data_destructor)
QM_STUB(cuda_data_in)
QM_STUB(cuda_dense_bwd)
QM_STUB(cuda_dense_constructor)
QM_STUB(cuda_dense_destructor)
QM_STUB(cuda_dense_fwd)
QM_STUB(cuda_dense_init)
QM_STUB(cuda_dense_mat_bwd)
QM_STUB(cuda_dense_mat_constructor)
QM_STUB(cuda_dense_mat_destructor)
QM_STUB(cuda_dense_mat_fwd)
QM_STUB(cuda_dense_mat_init)
QM_STUB(cuda_dense_mat_w_up)
QM_STUB(cuda_dense_w_up)
QM_STUB(cuda_dot_mat_vec_bwd)
QM_STUB(cuda_dot_mat_vec_bwd_appx)
QM_STUB(cuda_dot_mat_vec_constructor)
QM_STUB(cuda_dot_mat_vec_destructor)
QM_STUB(cuda_dot_mat_vec_fwd)
QM_STUB(cuda_dot_mat_vec_fwd_appx)
QM_STUB(cuda_dot_mat_vec_init)
QM_STUB(cuda_dup_grad_bwd)
QM_STUB(cuda_dup_grad_constructor)
QM_STUB(cuda_dup_grad_destructor)
QM_STUB(cuda_mult_e_mat_bwd)
QM_STUB(cuda_mult_e_mat_constructor)
QM_STUB(cuda_mult_e_mat_destructor)
QM_STUB(cuda_mult_e_mat_fwd)
QM_STUB(cuda_mult_e_mat_init)
QM_STUB(cuda_mult_e_vec_bwd)
QM_STUB(cuda_mult_e_vec_constructor)
QM_STUB(cuda_mult_e_vec_destructor)
QM_STUB(cuda_mult_e_vec_fwd)
QM_STUB(cuda_mult_e_vec_init)
QM_STUB(cuda_scale_bwd)
QM_STUB(cuda_scale_constructor)
QM_STUB(cuda_scale_destructor)
QM_STUB(cuda_scale_fwd)
QM_STUB(cuda_scale_init)
QM_STUB(cuda_scale_w_up)
QM_STUB(cuda_set_value)
QM_STUB(cuda_softmax_bwd)
QM_STUB(cuda_softmax_constructor)
QM_STUB(cuda_softmax_destructor)
QM_STUB(cuda_softmax_fwd)
QM_STUB(cuda_softmax_init)
QM_STUB(cuda_sum_vec_bwd)
QM_STUB(cuda_sum_vec_constructor)
QM_STUB(cuda_sum_vec_destructor)
QM_STUB(cuda_sum_vec_fwd)
QM_STUB(cuda_sum_vec_init)
