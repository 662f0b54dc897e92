from objectdetectionpl_amd.callbacks import EarlyStopping  # noqa: F401
