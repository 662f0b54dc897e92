from objectdetectionpl_amd.loggers import TensorBoardLogger  # noqa: F401
