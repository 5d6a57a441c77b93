// Research code of csrc/attention.hip (section 4), compiled only with -DFTMI_EXPERIMENTAL (FTMI_EXPERIMENTAL=1 python -m finetrainers_amd.csrc.build):
// measured experiments kept with their results (profiles/r02_attention_experiments.txt, r03_attention_experiments.txt, r04_cross_attention.txt) -- NOT part of
// libftmi355.so.  Included textually inside namespace ftmi at the point of attention.hip where the section used to live.

    const int fv = env_int("FTMI_ATTN_FWD", 0);  // re-read every call: tools/bench_attn.py switches variants inside one process
    if (fv && !(a.kbias || (a.Sk % 64) != 0)) {
        switch (fv) {
#define FTMI_AF(id, FLAGS, MINW) case id: hipLaunchKernelGGL((attn_fwd_kernel<false, FLAGS, MINW>), grid, dim3(256), kFwdLds, st, a); return check_launch("attn_fwd");
            FTMI_AF(1, AF_VALU_ROWSUM, 1)
            FTMI_AF(2, AF_LAZY, 1)
            FTMI_AF(3, AF_VALU_ROWSUM | AF_LAZY, 1)
            FTMI_AF(4, AF_ABL_NOEXP, 1)
            FTMI_AF(5, AF_ABL_NOPV, 1)
            FTMI_AF(6, AF_ABL_NOLOAD, 1)
            FTMI_AF(7, AF_ABL_NOEXP | AF_ABL_NOLOAD, 1)
            FTMI_AF(8, AF_ABL_NOEXP | AF_ABL_NOPV | AF_ABL_NOLOAD, 1)
            FTMI_AF(10, 0, 2)
            FTMI_AF(11, 0, 4)
            FTMI_AF(12, AF_VALU_ROWSUM | AF_LAZY, 2)
            FTMI_AF(13, AF_VALU_ROWSUM | AF_LAZY, 4)
            FTMI_AF(14, AF_LAZY, 2)
            FTMI_AF(15, AF_LAZY, 4)
            FTMI_AF(16, AF_LAZY | AF_MAX16, 1)
            FTMI_AF(17, AF_MAX16, 1)
            FTMI_AF(18, AF_LAZY | AF_MAX16, 2)
            FTMI_AF(19, AF_LAZY | AF_MAX16 | AF_TIMING, 1)
#undef FTMI_AF
            default: break;
        }
    }
