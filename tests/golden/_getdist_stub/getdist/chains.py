class WeightedSampleError(Exception):
    pass


class WeightedSamples:
    pass
