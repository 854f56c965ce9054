class Data:  # placeholder
    pass


class InMemoryDataset:  # placeholder
    pass
