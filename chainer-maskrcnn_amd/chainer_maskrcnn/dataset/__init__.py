"""Host data pipeline: COCO loaders, the training transforms and a prefetching batch loader."""
