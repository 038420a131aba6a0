"""casapose_amd -- MI355X-native (gfx950) CASAPose hot path behind the reference's Python API.

    from casapose_amd.pose_models.tfkeras import Classifiers          # model registry
    from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted
    from casapose_amd.pose_estimation.ransac_voting import ransac_voting_layer_all_masks

Arithmetic runs in hand-written HIP kernels (casapose_amd/csrc -> libcasapose_hip.so,
C ABI in include/casapose_hip.h); PyTorch-ROCm only owns device memory and streams.
"""
__version__ = "0.1.0"
