/*
 *  FeatureSegmentationImpl.scala -- replacement of the reference's class of this name (novelty curve on the GPU,
 *  break bookkeeping in the library's host code).
 */
package de.sciss.strugatzki
package impl

import de.sciss.processor.impl.ProcessorImpl

import scala.collection.breakOut
import scala.concurrent.blocking
import scala.xml.XML

private[strugatzki] final class FeatureSegmentationImpl(val config: FeatureSegmentation.Config)
  extends FeatureSegmentation with ProcessorImpl[FeatureSegmentation.Product, FeatureSegmentation] {

  import FeatureSegmentation._

  protected def body(): Product = blocking {
    Native.withContexts(Native.devices.take(1)) { ctx =>
      val res = Native.run(this, p => { progress = p }, () => checkAborted()) { obs =>
        Native.runSegmentation(ctx(0), config.toXML.toString, obs)
      }
      (XML.loadString(res) \ "break").map(Break.fromXML)(breakOut): Product
    }
  }
}
