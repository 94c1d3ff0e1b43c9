"""Import-path mirror of the reference's `src/` package (inference.py:15,40-42): the classes live in idm_vton_amd.boundary."""
