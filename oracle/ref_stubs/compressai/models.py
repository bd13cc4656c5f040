class CompressionModel:
    pass
