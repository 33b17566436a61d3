"""Face detectors of the reference's `facenet.detectors` package that run on the MI355X path."""
