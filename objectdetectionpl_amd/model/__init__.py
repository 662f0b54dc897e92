"""Mirror of the reference's `model` package: model classes as callers of the hot path."""
