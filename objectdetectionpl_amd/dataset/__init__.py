"""Input side of the hot path: the batch wire format of the reference's data modules."""
