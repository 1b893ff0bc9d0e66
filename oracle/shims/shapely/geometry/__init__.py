class Point: pass
class Polygon: pass
class MultiPolygon: pass
