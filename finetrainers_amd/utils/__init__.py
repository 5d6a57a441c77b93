from .diffusion import (  # noqa: F401
    compute_density_for_timestep_sampling,
    compute_loss_weighting_for_sd3,
    default_flow_shift,
    prepare_loss_weights,
    prepare_sigmas,
)
