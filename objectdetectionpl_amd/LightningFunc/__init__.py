"""Mirror of the reference's `LightningFunc` package (hot-path modules only)."""
