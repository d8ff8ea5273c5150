class IterDataPipe: pass
class Concater: pass
class IterableWrapper: pass
class ZipperLongest: pass
class Zipper: pass
