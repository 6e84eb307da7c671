"""Label vocabularies of the reference's COCO-Gen evaluation (``daam/experiment.py:17-92``): the 80 COCO "thing" names in
category-id order, the 27 coarse COCO-Stuff names, the small hypernym ontology used to build word lists, and the fold of the
80 names onto the coarse ones that ``simplify80=True`` applies to mask names.  Data only; kept as compact text tables and
expanded at import."""
from __future__ import annotations

from typing import Dict, List

__all__ = ['COCO80_LABELS', 'COCO80_INDICES', 'COCOSTUFF27_LABELS', 'COCO80_ONTOLOGY', 'COCO80_TO_27', 'UNUSED_LABELS',
           'build_word_list_coco80']


def _names(text: str) -> List[str]:
    return [name.strip() for name in text.replace('\n', ' ').split(',') if name.strip()]


COCO80_LABELS: List[str] = _names('''
    person, bicycle, car, motorcycle, airplane, bus, train, truck, boat, traffic light, fire hydrant, stop sign,
    parking meter, bench, bird, cat, dog, horse, sheep, cow, elephant, bear, zebra, giraffe, backpack, umbrella, handbag,
    tie, suitcase, frisbee, skis, snowboard, sports ball, kite, baseball bat, baseball glove, skateboard, surfboard,
    tennis racket, bottle, wine glass, cup, fork, knife, spoon, bowl, banana, apple, sandwich, orange, broccoli, carrot,
    hot dog, pizza, donut, cake, chair, couch, potted plant, bed, dining table, toilet, tv, laptop, mouse, remote, keyboard,
    cell phone, microwave, oven, toaster, sink, refrigerator, book, clock, vase, scissors, teddy bear, hair drier, toothbrush
''')
COCO80_INDICES: Dict[str, int] = {name: i for i, name in enumerate(COCO80_LABELS)}

# placeholders a composite prediction image is decoded with when no vocabulary is given (experiment.py:33,179-180)
UNUSED_LABELS: List[str] = [f'__unused_{i}__' for i in range(1, 200)]

# (the reference's list repeats 'food' and 'furniture': indices are positions in THIS list, so the repeats stay)
COCOSTUFF27_LABELS: List[str] = _names('''
    electronic, appliance, food, furniture, indoor, kitchen, accessory, animal, outdoor, person, sports, vehicle, ceiling,
    floor, food, furniture, rawmaterial, textile, wall, window, building, ground, plant, sky, solid, structural, water
''')

# hypernym -> members (members may be hypernyms themselves)
COCO80_ONTOLOGY: Dict[str, List[str]] = {
    head: _names(members) for head, members in (line.split(':') for line in '''
        two-wheeled vehicle: bicycle, motorcycle
        vehicle: two-wheeled vehicle, four-wheeled vehicle
        four-wheeled vehicle: bus, truck, car
        four-legged animals: livestock, pets, wild animals
        livestock: cow, horse, sheep
        pets: cat, dog
        wild animals: elephant, bear, zebra, giraffe
        bags: backpack, handbag, suitcase
        sports boards: snowboard, surfboard, skateboard
        utensils: fork, knife, spoon
        receptacles: bowl, cup
        fruits: banana, apple, orange
        foods: fruits, meals, desserts
        meals: sandwich, hot dog, pizza
        desserts: cake, donut
        furniture: chair, couch, bench
        electronics: monitors, appliances
        monitors: tv, cell phone, laptop
        appliances: oven, toaster, refrigerator
    '''.strip().splitlines())
}
COCO80_ONTOLOGY = {head.strip(): members for head, members in COCO80_ONTOLOGY.items()}

# coarse name <- the COCO-80 names folded onto it ('person' has no entry: it keeps its name)
_COARSE_MEMBERS = {
    'vehicle': 'bicycle, car, motorcycle, airplane, bus, train, truck, boat',
    'accessory': 'traffic light, fire hydrant, stop sign, parking meter, backpack, umbrella, handbag, tie, suitcase',
    'furniture': 'bench, chair, couch, bed, dining table, toilet',
    'animal': 'bird, cat, dog, horse, sheep, cow, elephant, bear, zebra, giraffe',
    'sports': 'frisbee, skis, snowboard, sports ball, kite, baseball bat, baseball glove, skateboard, surfboard, tennis racket',
    'food': 'bottle, wine glass, cup, fork, knife, spoon, bowl, banana, apple, sandwich, orange, broccoli, carrot, hot dog, '
            'pizza, donut, cake',
    'plant': 'potted plant',
    'electronic': 'tv, laptop, mouse, remote, keyboard, cell phone',
    'appliance': 'microwave, oven, toaster, sink, refrigerator',
    'indoor': 'book, clock, vase, scissors, teddy bear, hair drier, toothbrush',
}
COCO80_TO_27: Dict[str, str] = {name: coarse for coarse, members in _COARSE_MEMBERS.items() for name in _names(members)}


def build_word_list_coco80() -> Dict[str, List[str]]:
    """The ontology's LEAF groups: hypernyms none of whose members is itself a hypernym (experiment.py:87-91)."""
    return {head: members for head, members in COCO80_ONTOLOGY.items()
            if not any(member in COCO80_ONTOLOGY for member in members)}
