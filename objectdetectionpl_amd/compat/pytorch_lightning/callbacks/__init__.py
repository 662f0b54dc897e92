from objectdetectionpl_amd.callbacks import (Callback, EarlyStopping, GPUStatsMonitor,  # noqa: F401
                                              LearningRateMonitor, ModelCheckpoint)
