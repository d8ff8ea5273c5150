class MapDataPipe: pass
