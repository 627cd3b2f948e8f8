from . import occ_grid, prop_net  # noqa: F401
