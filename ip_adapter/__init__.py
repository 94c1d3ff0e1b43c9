"""Import-path mirror of the reference's `ip_adapter/` package (plugin API: attention processors and the Resampler)."""
