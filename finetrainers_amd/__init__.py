"""finetrainers_amd -- MI355X (gfx950) native backend for the finetrainers LTX-Video LoRA SFT step.

Only what the hot path needs lives here: ``csrc/`` (hand-written HIP kernels + the C ABI of
include/ftmi355.h), the ctypes binding, and the host-side mirrors of the reference's two plugin surfaces
(``ModelSpecification`` and the attention-provider registry) plus the step/DP glue of SFTTrainer._train.
"""

__version__ = "0.1.0"
