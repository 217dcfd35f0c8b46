"""Host-side helpers of the package; `transformations` mirrors the six functions of
abr_control/utils/transformations.py that the control path uses (evaluated on the GPU)."""
