"""Drop-in import name of the reference's simple-knn extension package (imported as
`from simple_knn._C import distCUDA2`, /root/reference/scene/gaussian_model.py:19)."""
