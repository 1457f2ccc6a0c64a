"""Drop-in counterparts of the reference's `utils` hot-path modules."""
