"""dataset/AdvData_dataset.py mirror: read_PC / AdvData_Dataset (see cloud_io.py)."""
from .cloud_io import AdvData_Dataset, read_PC  # noqa: F401
