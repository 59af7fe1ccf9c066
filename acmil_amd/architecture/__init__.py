"""Mirror of the reference's `architecture` package for the hot path: same module names, constructor
signatures, forward() signatures and state_dict keys."""
