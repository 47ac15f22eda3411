// placeholder until the LM solver lands
