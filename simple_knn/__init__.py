"""Import-name shim for the un-vendored simple-knn submodule (r2_gaussian/gaussian/gaussian_model.py:21:
``from simple_knn._C import distCUDA2``)."""
