"""dataset/bosphorus_dataset.py — text-cloud part only (the `.bnt` scanner format and the csv index of the private
Bosphorus database are out of scope: SURVEY §2.1); see cloud_io.py."""
from .cloud_io import load_cloud_txt, normalize_cloud, rand_row  # noqa: F401
