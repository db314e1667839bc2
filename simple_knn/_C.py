from r2_gaussian_amd._C import distCUDA2   # noqa: F401
