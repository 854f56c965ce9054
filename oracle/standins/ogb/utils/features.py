ATOM_FEATURE_DIMS = [119, 4, 12, 12, 10, 6, 6, 2, 2]
BOND_FEATURE_DIMS = [5, 6, 2]


def get_atom_feature_dims():
    return list(ATOM_FEATURE_DIMS)


def get_bond_feature_dims():
    return list(BOND_FEATURE_DIMS)
